#!/usr/bin/env python
"""Benchmark of LeftRefill's diffusion-sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...

One "step" = one pass of the hot path over one batch: a full DDIM sampling run (S=50 steps, classifier-free guidance
2.5, eta=1) of B=4 stitched 512x1024 canvases (latent 64x128, UNet batch 2B=8, fp16) through the drop-in entry point
`inpainting_ldm.ref_inpainting_ldm.RefInpaintLDM.sample_log` -> `DDIMSampler.sample` -> hipGraph-replayed UNet step +
fused CFG/DDIM update.  That is BASELINE.json configs[1].  With N GPUs every rank samples its own B=4 batch (weak
scaling, no collective on the data path -- SURVEY.md section 8e); value = all images of all ranks / wall time.

Synthetic data (no dataset, no SD2 weights in the reference): seeded normal weights with fan-in scaling, x_T ~ N(0,1),
c_concat = [blocky right-half mask | 4 latent channels ~ N(0, 0.18215^2)], contexts ~ N(0,1) [77,1024].  The VAE and
prompt encoder are outside the path (host PyTorch code) and outside the timed region.

The printed JSON line also carries
  roofline     : the dominant kernel family (implicit-GEMM conv/linear `gemm_conv_kernel`) timed live with HIP events on
                 its launch stream in one instrumented eager UNet step: achieved = algorithmic FLOPs / sum of durations;
  cpu_baseline : the CPU oracle (oracle/, fp32 torch, this box's host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UNET_PARAMS = dict(use_checkpoint=True, image_size=32, in_channels=9, out_channels=4, model_channels=320,
                   attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64,
                   use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
                   legacy=False)  # check_points/ref_guided_inpainting/model_config.yaml:20-36
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
S_DDIM, CFG, ETA = 50, 2.5, 1.0


# config 4: view_num / concat_target and the canvas size of one view, all canvases of a sample on ONE GPU
MV_WORKLOADS = {"mv5": dict(view_num=5, concat_target=True, h=64, w=128),     # 4 canvases [ref_i | target], seq 5*4096
                "mv4": dict(view_num=4, concat_target=False, h=64, w=64)}     # 4 square views, seq 4*4096


def build_model(device, workload="single"):
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM
    target, params = "ldm.modules.diffusionmodules.openaimodel.UNetModel", dict(UNET_PARAMS)
    if workload == "nvs":       # BASELINE configs[4]: the NVS task model (configs/nvs_training_config.yaml: plain UNetModel, no separator tokens)
        from inpainting_ldm.NVS_ldm import NVSLDM
        model = NVSLDM(first_stage_config={"target": "torch.nn.Identity"}, cond_stage_config={"target": "torch.nn.Identity"},
                       unet_config={"target": target, "params": params}, conditioning_key="hybrid", scale_factor=0.18215,
                       linear_start=0.00085, linear_end=0.0120, timesteps=1000, channels=4, image_size=64, first_stage_key="image",
                       cond_stage_key="txt", data_config={"img_size": 256})
        return _init_weights(model, device)
    if workload != "single":
        mv = MV_WORKLOADS[workload]
        target = "ldm.modules.diffusionmodules.multiview_unet.MultiViewUnetModel"
        params.update(view_num=mv["view_num"], concat_target=mv["concat_target"])
    model = RefInpaintLDM(first_stage_config={"target": "torch.nn.Identity"},
                          cond_stage_config={"target": "torch.nn.Identity"},
                          unet_config={"target": target, "params": params},
                          conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                          timesteps=1000, channels=4, image_size=64, first_stage_key="image", cond_stage_key="txt",
                          data_config={"img_size": 512})
    return _init_weights(model, device)


def _init_weights(model, device):
    model = model.to(device).eval()
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for name, p in model.model.diffusion_model.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, device=device, generator=g) * (1.0 / fan_in) ** 0.5)
            elif name.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, device=device, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, device=device, generator=g))
    model.model.diffusion_model.prepare()
    return model


def synthetic_batch(B, h, w, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    mask = torch.zeros(B, 1, h, w, device=device)
    blocks = (torch.rand(B, 1, h // 8, w // 16, device=device, generator=g) < 0.5).float()
    mask[:, :, :, w // 2:] = torch.nn.functional.interpolate(blocks, size=(h, w // 2), mode="nearest")
    lat = torch.randn(B, 4, h, w, device=device, generator=g) * 0.18215
    c_concat = torch.cat([mask, lat], dim=1)
    c_cross = torch.randn(B, 77, 1024, device=device, generator=g)
    uc_cross = torch.randn(B, 77, 1024, device=device, generator=g)
    x_T = torch.randn(B, 4, h, w, device=device, generator=g)
    return c_concat, c_cross, uc_cross, x_T


def sample_once(model, batch, B, steps=None):
    steps = steps or S_DDIM
    c_concat, c_cross, uc_cross, x_T = batch
    cond = {"c_concat": [c_concat], "c_crossattn": [c_cross]}
    uc = {"c_concat": [c_concat], "c_crossattn": [uc_cross]}
    samples, _ = model.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=steps, eta=ETA,
                                  unconditional_guidance_scale=CFG, unconditional_conditioning=uc, x_T=x_T)
    return samples


def eager_unet_step(unet, x, t, ctx, hook=None):
    """One UNet forward with the launches of the captured step, eagerly: the cross-attention K / V projections of the
    (constant) context come from `_context_kv` like in the hipGraph path -- computed BEFORE `hook()` arms the instrumentation,
    so the measured launches are exactly those a DDIM step replays."""
    unet.prepare()
    x = x.float().contiguous()
    t = t.to(torch.int64).contiguous()
    ctx = ctx.to(unet.compute_dtype).contiguous()
    kv = unet._context_kv(ctx)
    if hook is not None:
        hook()
    return unet._run_plan(x, t, ctx, kv)


def kernel_roofline(model, batch, B, dump=None):
    """One instrumented eager UNet step: HIP events around every gemm_conv / attention launch on the launch stream."""
    from leftrefill_amd import ops
    from leftrefill_amd.flops import unet_flops
    unet = model.model.diffusion_model
    c_concat, c_cross, uc_cross, x_T = batch
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
    t = torch.full((2 * B,), 501, device=x.device, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half()
    rec = {"gemm_conv": [], "attention": [], "xattn_block": [], "ffn_block": [], "stin_block": [], "rowlin": []}
    orig = {"gemm_conv": ops.gemm_conv, "attention": ops.attention, "xattn_block": ops.xattn_block, "ffn_block": ops.ffn_block,
            "stin_block": ops.stin_block, "rowlin": ops.rowlin}

    def wrap(name):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # torch.cuda.current_stream() == the stream the kernel is launched on (ops._stream)
            out = orig[name](*a, **k)
            e1.record()
            if name == "gemm_conv":
                desc = dict(M=k["B"] * k["H"] * k["W"], N=a[1].shape[-2], K=a[1].shape[-1], taps=k.get("taps", 1),      # (per-sample weights: [B, N, K])
                            stride=k.get("stride", 1), up=k.get("up", 0), geglu=bool(k.get("geglu", False)),
                            cat=k.get("x2") is not None, resid=k.get("resid") is not None)
            elif name == "xattn_block":      # fused LayerNorm + to_q + 77-key attention + to_out + residual (level 0)
                desc = dict(M=a[0].shape[0], C=a[0].shape[1], Lc=k["Lc"], heads=k["heads"], pre=int(k.get("pre") is not None))
            elif name == "ffn_block":        # fused LayerNorm + GEGLU projection + gate + second Linear + residual (level 0)
                desc = dict(M=a[0].shape[0], C=a[0].shape[1], H=a[3].shape[0] * 64)
            elif name == "stin_block":       # fused proj_in + LayerNorm + q|k|v projection (level 0)
                desc = dict(M=a[0].shape[0], C=a[0].shape[1], NQ=a[3].shape[0])
            elif name == "rowlin":           # row-resident LayerNorm + Linear (level 1: q|k|v, GEGLU)
                desc = dict(M=a[0].shape[0], C=a[0].shape[1], N=a[1].shape[0], geglu=int(bool(k.get("geglu", False))))
            else:
                desc = dict(B=a[3], heads=a[4], Nq=a[5], Nkv=a[6])
            rec[name].append((e0, e1, desc))
            return out
        return f

    unet.use_hip_graph = False
    try:
        def arm():
            for k in rec:
                rec[k].clear()
            for n_ in rec:
                setattr(ops, n_, wrap(n_))
        for _ in range(2):
            for n_ in rec:
                setattr(ops, n_, orig[n_])
            with torch.no_grad():
                eager_unet_step(unet, x, t, ctx, hook=arm)
            torch.cuda.synchronize()
    finally:
        for n_ in rec:
            setattr(ops, n_, orig[n_])
        unet.use_hip_graph = True
    fl = unet_flops(unet, x.shape[2], x.shape[3])
    n = 2 * B
    # algorithmic HBM bytes of the GEMM family: every source element, weight and residual read once, output written once
    gbytes = 0.0
    for _, _, d in rec["gemm_conv"]:
        src_rows = d["M"] * (4 if d["stride"] == 2 else 1) / (4 if d["up"] else 1)
        n_out = d["N"] // 2 if d["geglu"] else d["N"]
        gbytes += 2.0 * (src_rows * d["K"] / d["taps"] + d["N"] * d["K"] + d["M"] * n_out * (2 if d["resid"] else 1))
    out = {}
    # the fused cross-attention block runs two of the UNet's pointwise linears and a 77-key attention: its flops leave the
    # numerators of the GEMM / attention families (which only count what those kernels still execute)
    ffn_fl = lambda d: 6.0 * d["M"] * d["C"] * d["H"]      # [M, C] x [C, 2H] + [M, H] x [H, C]
    xl_fl = lambda d: (4.0 + 2.0 * d["pre"]) * d["M"] * d["C"] * d["C"]      # to_q + to_out (+ the self-attention's out-projection)
    stin_fl = lambda d: 2.0 * d["M"] * d["C"] * (d["C"] + d["NQ"])      # proj_in + the fused q|k|v projection
    moved = {"gemm": sum(xl_fl(d) for _, _, d in rec["xattn_block"]) + sum(ffn_fl(d) for _, _, d in rec["ffn_block"])
                     + sum(stin_fl(d) for _, _, d in rec["stin_block"]) + sum(2.0 * d["M"] * d["C"] * d["N"] for _, _, d in rec["rowlin"]),
             "attn": sum(4.0 * d["M"] * d["Lc"] * d["C"] for _, _, d in rec["xattn_block"])}
    for name, key in (("gemm_conv", "gemm"), ("attention", "attn")):
        ms = sum(a.elapsed_time(b) for a, b, _ in rec[name])
        out[name] = {"launches": len(rec[name]), "total_ms": ms, "avg_us": 1e3 * ms / max(1, len(rec[name])),
                     "tflops": (n * fl[key] - moved[key]) / (ms * 1e-3) / 1e12}
    if rec["xattn_block"]:
        ms = sum(a.elapsed_time(b) for a, b, _ in rec["xattn_block"])
        fl_x = sum(xl_fl(d) + 4.0 * d["M"] * d["Lc"] * d["C"] for _, _, d in rec["xattn_block"])
        out["xattn_block"] = {"launches": len(rec["xattn_block"]), "total_ms": ms, "avg_us": 1e3 * ms / len(rec["xattn_block"]),
                              "tflops": fl_x / (ms * 1e-3) / 1e12}
    if rec["ffn_block"]:
        ms = sum(a.elapsed_time(b) for a, b, _ in rec["ffn_block"])
        out["ffn_block"] = {"launches": len(rec["ffn_block"]), "total_ms": ms, "avg_us": 1e3 * ms / len(rec["ffn_block"]),
                            "tflops": sum(ffn_fl(d) for _, _, d in rec["ffn_block"]) / (ms * 1e-3) / 1e12}
    if rec["stin_block"]:
        ms = sum(a.elapsed_time(b) for a, b, _ in rec["stin_block"])
        out["stin_block"] = {"launches": len(rec["stin_block"]), "total_ms": ms, "avg_us": 1e3 * ms / len(rec["stin_block"]),
                             "tflops": sum(stin_fl(d) for _, _, d in rec["stin_block"]) / (ms * 1e-3) / 1e12}
    if rec["rowlin"]:
        ms = sum(a.elapsed_time(b) for a, b, _ in rec["rowlin"])
        out["rowlin"] = {"launches": len(rec["rowlin"]), "total_ms": ms, "avg_us": 1e3 * ms / len(rec["rowlin"]),
                         "tflops": sum(2.0 * d["M"] * d["C"] * d["N"] for _, _, d in rec["rowlin"]) / (ms * 1e-3) / 1e12}
    out["gemm_conv"]["algorithmic_bytes_per_launch"] = gbytes / max(1, len(rec["gemm_conv"]))
    out["gemm_conv"]["algorithmic_gflop"] = (n * fl["gemm"] - moved["gemm"]) / 1e9
    # per-shape table of this instrumented step (what tools/kernel_table.py prints from --dump-kernels)
    agg = {}
    for name in rec:
        for a, b, d in rec[name]:
            us = 1e3 * a.elapsed_time(b)
            # by_ = algorithmic HBM bytes of the launch: every operand read once, every output written once (16-bit elements)
            if name == "gemm_conv":
                key = f'gemm {d["M"]}x{d["N"]}x{d["K"]} taps{d["taps"]} s{d["stride"]} up{d["up"]}' + (" geglu" if d["geglu"] else "") + (" cat" if d["cat"] else "")
                fl_ = 2.0 * d["M"] * d["N"] * d["K"]
                src_rows = d["M"] * (4 if d["stride"] == 2 else 1) / (4 if d["up"] else 1)
                by_ = 2.0 * (src_rows * d["K"] / d["taps"] + d["N"] * d["K"] + d["M"] * (d["N"] // 2 if d["geglu"] else d["N"]) * (2 if d["resid"] else 1))
            elif name == "xattn_block":
                key = f'xattn {d["M"]}x{d["C"]} keys{d["Lc"]} (' + ("attn1.to_out + resid + " if d["pre"] else "") + 'ln + to_q + attention + to_out + resid)'
                fl_ = (4.0 + 2.0 * d["pre"]) * d["M"] * d["C"] * d["C"] + 4.0 * d["M"] * d["Lc"] * d["C"]
                by_ = 2.0 * d["M"] * d["C"] * (2 + d["pre"]) + 2.0 * (2 + d["pre"]) * d["C"] * d["C"]
            elif name == "ffn_block":
                key = f'ffn {d["M"]}x{d["C"]} hidden{d["H"]} (ln + geglu proj + gate + linear + resid)'
                fl_ = 6.0 * d["M"] * d["C"] * d["H"]
                by_ = 2.0 * d["M"] * d["C"] * 3 + 2.0 * 3 * d["C"] * d["H"]
            elif name == "stin_block":
                key = f'stin {d["M"]}x{d["C"]} qkv{d["NQ"]} (proj_in + ln + q|k|v projection)'
                fl_ = stin_fl(d)
                by_ = 2.0 * d["M"] * (2 * d["C"] + d["NQ"]) + 2.0 * d["C"] * (d["C"] + d["NQ"])
            elif name == "rowlin":
                key = f'rowlin {d["M"]}x{d["N"]}x{d["C"]} (ln + linear' + (" + geglu gate)" if d["geglu"] else ")")
                fl_ = 2.0 * d["M"] * d["C"] * d["N"]
                by_ = 2.0 * d["M"] * (d["C"] + (d["N"] // 2 if d["geglu"] else d["N"])) + 2.0 * d["C"] * d["N"]
            else:
                key = f'attn B{d["B"]} h{d["heads"]} {d["Nq"]}x{d["Nkv"]}'
                fl_ = 4.0 * d["B"] * d["heads"] * d["Nq"] * d["Nkv"] * 64
                by_ = 2.0 * d["B"] * d["heads"] * 64 * (2 * d["Nq"] + 2 * d["Nkv"])
            e = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += us; e[2] += fl_; e[3] += by_
    out["table"] = [dict(shape=k, n=v[0], total_us=round(v[1], 1), avg_us=round(v[1] / v[0], 1), tflops=round(v[2] / v[1] / 1e6, 1),
                         gflop=round(v[2] / v[0] / 1e9, 3), alg_mb=round(v[3] / v[0] / 1e6, 2))
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    if dump:
        rows = []
        for name in rec:
            for a, b, d in rec[name]:
                us = 1e3 * a.elapsed_time(b)
                if name == "gemm_conv":
                    fl_ = 2.0 * d["M"] * d["N"] * d["K"]
                elif name == "xattn_block":
                    fl_ = (4.0 + 2.0 * d["pre"]) * d["M"] * d["C"] * d["C"] + 4.0 * d["M"] * d["Lc"] * d["C"]
                elif name == "ffn_block":
                    fl_ = 6.0 * d["M"] * d["C"] * d["H"]
                elif name == "stin_block":
                    fl_ = stin_fl(d)
                elif name == "rowlin":
                    fl_ = 2.0 * d["M"] * d["C"] * d["N"]
                else:
                    fl_ = 4.0 * d["B"] * d["heads"] * d["Nq"] * d["Nkv"] * 64
                rows.append(dict(kernel=name, us=us, tflops=fl_ / us / 1e6, **d))
        with open(dump, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
    return out, fl


def unet_step_events(model, batch, B, n=25, warm=5):
    """SURVEY 8d reporting: HIP-event time of ONE graph-replayed UNet forward at batch 2B (median of n after warm),
    and of the fused CFG + DDIM update, separately."""
    from leftrefill_amd import ops
    unet = model.model.diffusion_model
    c_concat, c_cross, uc_cross, x_T = batch
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
    t = torch.full((2 * B,), 501, device=x.device, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half()
    evs = []
    with torch.no_grad():
        for i in range(warm + n):        # no host sync inside the loop: launches stay queued ahead of the GPU like in sample()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            eps = unet(x, t, ctx)
            e1.record()
            ops.ddim_cfg_step(x_T, eps, x_T, CFG, 0.5, 0.6, 0.1, 0.7)
            e2.record()
            evs.append((e0, e1, e2))
    torch.cuda.synchronize()
    times = [a.elapsed_time(b) for a, b, _ in evs[warm:]]
    upd = [1e3 * b.elapsed_time(c) for _, b, c in evs[warm:]]
    times.sort()
    upd.sort()
    return {"unet_forward_ms_median": times[len(times) // 2], "unet_forward_ms_min": times[0],
            "ddim_update_us_median": upd[len(upd) // 2], "n": n, "warmup": warm, "unet_batch": 2 * B}


def vae_timing(B, device):
    """Next row, reported beside the metric, never inside `value`: KL-VAE decode of the B sampled latents
    and encode of B 512x1024 images on the same HIP kernels (shipped width, random-init weights)."""
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)     # model_config.yaml:44-58
    vae = AutoencoderKL(dd, {"target": "torch.nn.Identity"}, 4).to(device).eval()
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for p in vae.parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, device=device, generator=g) * (1.0 / p[0].numel()) ** 0.5)
    z = torch.randn(B, 4, 64, 128, device=device, generator=g)
    x = torch.randn(B, 3, 512, 1024, device=device, generator=g).clamp(-1, 1)
    out = {}
    for name, fn, tflop in (("decode", lambda: vae.decode(z), 5.10), ("encode", lambda: vae.encode(x), 2.30)):
        fn()                                  # tile autotune + weight packing
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out[name] = {"ms": ms, "batch": B, "tflops": B * tflop / (ms * 1e-3)}
    return out


def train_bench(a, rank, world, device, model=None, steps=None):
    """`--workload train` (not the metric; SURVEY 8f-2 / BASELINE configs[4]): one optimisation step of the prompt tokens
    through the frozen UNet at canvas 256x512 (latent 32x64), per-GPU batch 16: p_losses forward + HIP backward to the
    context + AdamW on the tokens (stand-in for `special_embeddings`, 73 x 1024), loss scale 2^14, data-parallel
    all-reduce of the token gradient over RCCL when world > 1."""
    import torch.distributed as dist
    from leftrefill_amd import dist as lrd
    Bt, h, w = 16, 32, 64
    steps = steps or a.steps
    task = getattr(a, "task", "nvs")
    model = (model or build_model(device, "nvs" if task == "nvs" else "single")).train()
    bf16 = getattr(a, "dtype", "f16") == "bf16"
    unet = model.model.diffusion_model
    prev_dtype = unet.compute_dtype
    unet.compute_dtype = torch.bfloat16 if bf16 else torch.float16      # re-packs the weights on the next forward
    unet.recompute_in_backward = bool(getattr(a, "recompute", False))
    for p in model.parameters():
        p.requires_grad_(False)
    g0 = torch.Generator(device=device).manual_seed(99)            # parameters: the SAME initial tokens on every rank
    tokens = torch.nn.Parameter(0.02 * torch.randn(73, 1024, device=device, generator=g0))
    g = torch.Generator(device=device).manual_seed(1099 + rank)     # data: rank-dependent
    trainable = [tokens]
    pose_mlp = rel_pos = None
    if task == "nvs":
        # the NVS prompt encoder's pose token (ldm/modules/encoders/NVS_modules.py:92-106, 219-224): RelPosModel(4 -> 512 -> 1024) of the
        # relative camera pose, trained with the prompt tokens (NVS_ldm.py:326-329); here it is added at its token slot of the context
        # (the CLIP tower between the token embeddings and the context needs the absent open_clip weights)
        import leftrefill_amd.dropin as dropin
        dropin.install()
        from ldm.modules.encoders.NVS_modules import RelPosModel
        torch.manual_seed(7)                                        # same initial pose MLP on every rank
        pose_mlp = RelPosModel(input_ch=4, out_ch=1024).to(device)
        rel_pos = torch.randn(Bt, 4, device=device, generator=g)
        trainable += list(pose_mlp.parameters())
    opt = torch.optim.AdamW(trainable, lr=1e-4)
    base_ctx = torch.randn(Bt, 77, 1024, device=device, generator=g)
    c_concat = torch.randn(Bt, 5, h, w, device=device, generator=g)
    x_start = torch.randn(Bt, 4, h, w, device=device, generator=g)
    # dynamic loss scale like the reference's fp16 AMP (Lightning precision=16 -> GradScaler, train_inpainting.py:52,127):
    # a non-finite scaled gradient skips the step and halves the scale, 200 clean steps double it.  The hipGraph variant
    # keeps the scale fixed (the decision needs the host).
    # bf16 (BASELINE configs[4]) has fp32's exponent range: no loss scale.
    scaler = {"scale": 1.0 if bf16 else 2.0 ** 14, "good": 0, "skipped": 0}

    t_buf = torch.zeros(Bt, device=device, dtype=torch.long)
    noise_buf = torch.zeros(Bt, 4, h, w, device=device)

    def draw():
        t_buf.copy_(torch.randint(0, 1000, (Bt,), device=device, generator=g))
        noise_buf.copy_(torch.randn(Bt, 4, h, w, device=device, generator=g))

    def body():
        ctx = torch.cat([base_ctx[:, :1], base_ctx[:, 1:74] + tokens, base_ctx[:, 74:]], dim=1)
        if pose_mlp is not None:
            ctx = torch.cat([ctx[:, :74], ctx[:, 74:75] + pose_mlp(rel_pos)[:, None], ctx[:, 75:]], dim=1)
        loss, _ = model.p_losses(x_start, {"c_concat": [c_concat], "c_crossattn": [ctx]}, t_buf, noise=noise_buf)
        (loss * scaler["scale"]).backward()
        lrd.allreduce_mean_grads(trainable)          # after the reduction every rank sees the same gradient -> same decision
        if not torch.cuda.is_current_stream_capturing() and not all(bool(torch.isfinite(p_.grad).all()) for p_ in trainable):
            scaler["scale"] *= 0.5
            scaler["good"] = 0
            scaler["skipped"] += 1
            return loss
        for p_ in trainable:
            p_.grad /= scaler["scale"]
        opt.step()
        scaler["good"] += 1
        if scaler["good"] % 200 == 0 and not bf16:
            scaler["scale"] *= 2.0
        return loss

    graph = None
    if getattr(a, "train_graph", False):
        # whole step (forward, HIP backward, token all-reduce, AdamW) captured into ONE hipGraph; t / noise are refreshed
        # in place before every replay
        opt = torch.optim.AdamW(trainable, lr=1e-4, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                draw()
                opt.zero_grad(set_to_none=True)
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = body()

    def step():
        draw()
        if graph is not None:
            graph.replay()
            return static_loss
        loss = body()
        opt.zero_grad(set_to_none=True)
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, a.warmup)):
        loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(loss)
    peak_gb = torch.cuda.max_memory_allocated(device) / 2 ** 30
    model.eval()
    with torch.no_grad():      # forward alone, same shapes (eager, like the training forward)
        unet.use_hip_graph = False
        xin = torch.cat([x_start, c_concat], 1)
        tt_ = torch.full((Bt,), 501, device=device, dtype=torch.long)
        unet(xin, tt_, base_ctx.to(unet.compute_dtype))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            unet(xin, tt_, base_ctx.to(unet.compute_dtype))
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) / 3 * 1e3
    unet.compute_dtype = prev_dtype
    from leftrefill_amd.flops import unet_train_flops
    tf = unet_train_flops(unet, h, w)
    step_tflops = Bt * tf["total"] / (dt / steps) / 1e12
    roof = {"bound": "mfma", "kernel": "whole training step (forward + input-gradient backward; AdamW on 73 x 1024 tokens is negligible)",
            "achieved": step_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": step_tflops / MFMA_PEAK_TFLOPS,
            "algorithmic_tflop_per_step": Bt * tf["total"] / 1e12, "forward_tflop": Bt * tf["forward"] / 1e12,
            "backward_tflop": Bt * tf["backward"] / 1e12,
            "forward_only_frac": Bt * tf["forward"] / (fwd_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
            "note": "2 MAC of the conv / linear / attention products of ONE step at batch 16 (leftrefill_amd/flops.py::unet_train_flops: dgrad "
                    "GEMMs for every layer behind the first cross-attention, attention backward = 2.5 x forward) / measured step time"}
    return {"metric": "training samples/sec (UNet fwd + bwd to the prompt tokens, frozen weights)", "value": world * Bt * steps / dt,
            "roofline": roof,
            "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f16", "data": "synthetic",
            "config": {"workload": ("configs[4] (NVS task model: NVSLDM.p_losses, prompt tokens + pose MLP trainable): " if task == "nvs"
                                    else "configs[4]-like (RefInpaintLDM): ") + "canvas 256x512 (latent 32x64), per-GPU batch 16, "
                                   + ("bf16 (no loss scale), " if bf16 else "fp16 + dynamic loss scale, ") +
                                   
                                   "p_losses + backward + AdamW on 73x1024 prompt tokens" + (" + RelPosModel (4-512-1024)" if task == "nvs" else ""),
                       "global_batch": world * Bt, "task": task,
                       "per_gpu_batch": Bt, "parallelism": f"dp{world} (all-reduce of the trainable token / pose-MLP gradients only)"},
            "forward_only_ms": fwd_ms, "final_loss": float(loss.detach()), "peak_memory_gib": peak_gb,
            "recompute_in_backward": bool(getattr(a, "recompute", False)), "hip_graph": graph is not None,
            "loss_scale": scaler["scale"], "skipped_steps": scaler["skipped"]}


def host_cpu():
    """(model name, physical cores, logical CPUs) of this box from /proc/cpuinfo (stated beside every CPU figure)."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
                phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or logical or (os.cpu_count() or 1)), logical or (os.cpu_count() or 1)


def cpu_baseline():
    """Oracle (fp32 torch CPU restatement of the reference, oracle/unet_ref.py + oracle/ddim_ref.py) on the host cores, SURVEY 8d.
    (1) thread sweep: one CFG UNet step (batch 2) at latent 32x64 for 16 / 32 / 64 / 128 torch threads (capped at the physical
    cores) -- all cores are NOT the fastest on a 128-core host; (2) with the fastest count: ONE CFG step of
    configs[1] (latent 64x128, batch 2) in both attention forms of the reference -- "naive" (CrossAttention.forward with
    materialised logits, attention.py:165-196) and "sdpa" (the fused form the reference runs with xformers, 199-250) -- `value` takes
    the FASTER of the two; (3) configs[0] in full through the oracle's sampler (latent 32x64, B = 1, 10 DDIM steps under CFG 2.5)
    with that count and form.  Every timing after one warm-up forward (thread pool, allocator)."""
    from oracle import ddim_ref, unet_ref
    cpu_model, phys, logical = host_cpu()
    prev_threads = torch.get_num_threads()
    prev_impl = unet_ref.ATTENTION_IMPL
    try:
        cfg = unet_ref.FULL
        g = torch.Generator().manual_seed(0)
        sd = {}
        for k, shp in unet_ref.param_shapes(cfg).items():
            if len(shp) >= 2:
                fan = 1
                for s_ in shp[1:]:
                    fan *= s_
                sd[k] = torch.randn(shp, generator=g) * (1.0 / fan) ** 0.5
            elif k.endswith(".weight"):
                sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
            else:
                sd[k] = 0.02 * torch.randn(shp, generator=g)
        ctx = torch.randn(2, 77, 1024, generator=g)
        t = torch.tensor([501, 501])
        x_small = torch.randn(2, 9, 32, 64, generator=g)
        x = torch.randn(2, 9, 64, 128, generator=g)

        def timed(fn):
            t0 = time.time()
            fn()
            return time.time() - t0

        torch.set_num_threads(max(1, min(phys, 32)))
        unet_ref.unet_forward(sd, cfg, x_small, t, ctx)      # warm-up (not timed)
        sweep = {}
        for nt in sorted({min(n_, max(1, phys)) for n_ in (16, 32, 64, 128)}):
            torch.set_num_threads(nt)
            unet_ref.unet_forward(sd, cfg, x_small[:, :, :8, :16], t, ctx)      # respawn the pool at this size (not timed)
            sweep[nt] = timed(lambda: unet_ref.unet_forward(sd, cfg, x_small, t, ctx))
        best_t = min(sweep, key=sweep.get)
        torch.set_num_threads(best_t)
        per_impl = {}
        for impl in ("naive", "sdpa"):
            unet_ref.ATTENTION_IMPL = impl
            per_impl[impl] = timed(lambda: unet_ref.unet_forward(sd, cfg, x, t, ctx))
        best_impl = min(per_impl, key=per_impl.get)
        unet_ref.ATTENTION_IMPL = best_impl
        dt = per_impl[best_impl]
        # configs[0] exactly as stated: the oracle's DDIM / CFG sampler, eta = 1 noise drawn per step like the reference
        x_T = torch.randn(1, 4, 32, 64, generator=g)
        c_concat = torch.randn(1, 5, 32, 64, generator=g)
        noises = [torch.randn(1, 4, 32, 64, generator=g) for _ in range(10)]
        dt0 = timed(lambda: ddim_ref.ddim_sample(lambda xc, tt, cc: unet_ref.unet_forward(sd, cfg, xc, tt, cc), 10, x_T, c_concat,
                                                 ctx[:1], ctx[1:], CFG, eta=ETA, noises=noises))
    finally:
        torch.set_num_threads(prev_threads)
        unet_ref.ATTENTION_IMPL = prev_impl
    sw = ", ".join(f"{n_} threads {v_:.2f} s" for n_, v_ in sorted(sweep.items()))
    return {"value": 1.0 / (50 * dt), "unit": "images/s", "cores": best_t, "kind": "port",
            "cpu_model": cpu_model, "physical_cores": phys, "logical_cpus": logical, "torch_threads": best_t,
            "thread_sweep_s_per_cfg_step_latent32x64": {str(k_): v_ for k_, v_ in sorted(sweep.items())},
            "s_per_unet_step_b2_naive": per_impl["naive"], "s_per_unet_step_b2_sdpa": per_impl["sdpa"], "attention_impl": best_impl,
            "images_per_s_naive": 1.0 / (50 * per_impl["naive"]), "images_per_s_sdpa": 1.0 / (50 * per_impl["sdpa"]),
            "sample": f"{cpu_model}, {phys} physical cores ({logical} logical); thread sweep on one CFG UNet step (batch 2) at latent "
                      f"32x64: {sw} -> {best_t} threads; configs[1]: 1 CFG UNet step (batch 2) at latent 64x128 with {best_t} threads: "
                      f"naive attention {per_impl['naive']:.2f} s, fused (SDPA) attention {per_impl['sdpa']:.2f} s -> `value` = the faster "
                      f"({best_impl}), extrapolated x50 steps per image; configs[0] in full through oracle/ddim_ref.ddim_sample (latent "
                      f"32x64, B=1, 10 DDIM steps, cfg 2.5, eta 1, {best_impl} attention, {best_t} threads) {dt0:.2f} s = {1.0 / dt0:.4f} images/s",
            "s_per_unet_step_b2": dt, "config0_full_s": dt0, "config0_images_per_s": 1.0 / dt0}


class HwSampler:
    """Shader clock and socket power of THIS rank's GPU, sampled from the amdgpu hwmon files (freq1_input = sclk in Hz,
    power1_input in microwatts) every `period` seconds by a background thread while the timed region runs.  The MI355X clocks to
    its power budget, so the dense-fp16 MFMA rate a kernel can be held against moves with the clock the step sustains
    (MI355X_MICROARCH.md, "DVFS give-back"); bench.py puts the measured mean next to the roofline fraction instead of quoting a
    clock from cycle-stamp arithmetic.  The hwmon directory is found through the PCI address of the torch device (the node's other
    GPUs are visible in sysfs too); fallback: one `rocm-smi --showclocks --showpower --json` poll per second."""

    def __init__(self, device, period=0.02):
        import glob
        import threading
        self.period, self.samples, self.src = period, [], None
        self._stop = threading.Event()
        self._thread = None
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(device)
            addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            cands = glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*")
            if cands and os.path.exists(os.path.join(cands[0], "freq1_input")):
                self.dir, self.src = cands[0], f"sysfs hwmon of {addr} (freq1_input / power1_input), {int(1 / period)} Hz"
        except Exception:      # noqa: BLE001
            self.dir = None
        if self.dir is None:
            import shutil
            self.smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
            if os.path.exists(self.smi):
                self.src, self.period = "rocm-smi --showclocks --showpower --json (card0), 1 Hz", 1.0

    def _read(self):
        if self.dir is not None:
            with open(os.path.join(self.dir, "freq1_input")) as f:
                mhz = int(f.read()) / 1e6
            with open(os.path.join(self.dir, "power1_input")) as f:
                w = int(f.read()) / 1e6
            return mhz, w
        import re
        import subprocess
        r = subprocess.run([self.smi, "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        js = json.loads(r.stdout[r.stdout.index("{"):])
        card = js[sorted(js)[0]]
        mhz = float(re.sub(r"[^0-9.]", "", card["sclk clock speed:"]))
        w = float([v for k, v in card.items() if "Power" in k][0])
        return mhz, w

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:      # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def start(self):
        if self.src is None:
            return self
        import threading
        self.samples = []
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=30)
            self._thread = None
        if not self.samples:
            return {"sclk_mhz_mean": None, "power_w_mean": None, "hw_sampler": self.src or "no hwmon files / rocm-smi on this box"}
        mhz = [m for m, _ in self.samples]
        w = [p_ for _, p_ in self.samples]
        return {"sclk_mhz_mean": sum(mhz) / len(mhz), "sclk_mhz_min": min(mhz), "sclk_mhz_max": max(mhz),
                "power_w_mean": sum(w) / len(w), "power_w_max": max(w), "hw_samples": len(mhz), "hw_sampler": self.src}


def sustained_mfma_peak():
    """tools/micro/mfma_peak.hip on this GPU, now: chip-wide register-only v_mfma_f32_16x16x32_f16 loops of >= 20 ms on random
    operands (two waves per SIMD, the GEMM's configuration) -> sustained TFLOP/s and the shader clock it ran at.  This is the
    dense-fp16 rate the matrix cores of THIS chip deliver inside its power budget; `roofline.peak` stays the datasheet constant."""
    import shutil
    import subprocess
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tools", "micro", "mfma_peak.hip")
    exe = "/tmp/lr_mfma_peak"
    if not os.path.exists(cc) or not os.path.exists(src):
        return {"error": "hipcc or tools/micro/mfma_peak.hip missing"}
    subprocess.run([cc, "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", exe, src], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    res = {"lines": [ln.strip() for ln in out.splitlines() if "operands" in ln]}
    import re
    for ln in res["lines"]:
        m = re.search(r"(random|zero)\s+operands\s+v_mfma_f32_(\S+)_f16\s+(\d) wave.*?([0-9.]+) TFLOP/s\s+shader clock\s+([0-9.]+) MHz", ln)
        if m and m.group(1) == "random" and m.group(3) == "2" and m.group(2) == "16x16x32":
            res["tflops_random_16x16x32"], res["sclk_mhz"] = float(m.group(4)), float(m.group(5))
        if m and m.group(1) == "random" and m.group(3) == "1" and m.group(2) == "32x32x16":
            res["tflops_random_32x32x16"] = float(m.group(4))
        if m and m.group(1) == "zero" and m.group(3) == "2" and m.group(2) == "16x16x32":
            res["tflops_zero_16x16x32"] = float(m.group(4))
    return res


def measure_traffic(launches):
    """HBM traffic of the GEMM family, measured NOW: tools/pmc_step.py (one eager UNet step at batch 8) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace domains), FETCH doubled per the gfx950
    correction of MI355X_MICROARCH.md.  Returns bytes per GEMM launch, or None when the profiler is not usable here."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="lr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            r = subprocess.run([prof, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, name), "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_step.py"), "run"], cwd=ROOT, env=env, capture_output=True,
                               text=True, timeout=600)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} failed: {r.stderr[-200:]}"
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_step
        out = os.path.join(tmp, "traffic.json")
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # stdout carries exactly one JSON line
            pmc_step.reduce_(os.path.join(tmp, "fetch"), os.path.join(tmp, "write"), out, launches)
        res = json.load(open(out))
        return res, None
    except Exception as e:      # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def emit_json_line(obj):
    """One JSON line on stdout as ONE write: several ranks may report at the same moment, and print() writes the text and the newline
    separately (two ranks' lines were seen glued together)."""
    sys.stdout.flush()
    os.write(1, (json.dumps(obj) + "\n").encode())


def dist_selftest(rank, world, device, backend, share):
    """First-contact checks of the multi-GPU job, before anything is timed: every stage is a named entry, the first
    failure is what the JSON line reports.  Semantics the collectives must reproduce: the all-gather before the re-arranged cross-view
    self-attention (reference ldm/modules/multiview_attention.py:436-462) and the sample sharding of configs[2]."""
    import torch.distributed as dist
    st = {}

    def stage(name, fn):
        try:
            if os.environ.get("LR_BENCH_SELFTEST_FAIL") == name:      # test hook: tests/test_gpu_bench.py checks the error line
                raise RuntimeError("forced failure (LR_BENCH_SELFTEST_FAIL)")
            st[name] = fn()
        except Exception as e:      # noqa: BLE001
            st[name] = {"ok": False, "error": f"{type(e).__name__}: {e}"[:300]}
        return bool(st[name].get("ok"))

    def world_size():
        return {"ok": dist.get_world_size() == world and dist.get_rank() == rank, "world": dist.get_world_size(), "backend": backend}

    def stamp():      # all-gather of a rank-stamped tensor: every slot must carry its rank, in rank order
        mine = torch.full((4,), float(rank), device=device)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        vals = [int(g[0].item()) for g in got]
        return {"ok": vals == list(range(world)) and all(bool((g == g[0]).all()) for g in got), "ranks_seen": vals}

    def into_tensor():      # the collective the sharded multi-view block uses (dist.mv_all_gather_rows): all_gather_into_tensor of fp16 rows
        mine = torch.full((8, 64), float(rank + 1), device=device, dtype=torch.float16)
        out = torch.empty(world * 8, 64, device=device, dtype=torch.float16)
        dist.all_gather_into_tensor(out, mine)
        want = torch.arange(1, world + 1, device=device, dtype=torch.float16).repeat_interleave(8)[:, None].expand(-1, 64)
        return {"ok": bool(torch.equal(out, want))}

    def reduce_sum():
        v = torch.tensor([float(rank + 1)], device=device, dtype=torch.float64)
        dist.all_reduce(v)
        return {"ok": abs(v.item() - world * (world + 1) / 2) < 1e-9, "sum": v.item()}

    def devices():      # one GPU per rank (unless the 1-GPU test hook shares cuda:0): no two ranks on the same device
        if device.type == "cuda":
            p = torch.cuda.get_device_properties(device)
            ident = f"{torch.cuda.current_device()}:{getattr(p, 'uuid', '')}:{getattr(p, 'pci_bus_id', '')}"
        else:      # (CPU ranks of tests/test_host_cpu.py)
            ident = f"cpu:{rank}"
        objs = [None] * world
        dist.all_gather_object(objs, ident)
        return {"ok": share or len(set(objs)) == world, "devices": objs, "shared_gpu_test_hook": bool(share)}

    for name, fn in (("world_size", world_size), ("all_gather_rank_stamp", stamp), ("all_gather_into_tensor_f16", into_tensor),
                     ("all_reduce_sum", reduce_sum), ("device_uniqueness", devices)):
        if not stage(name, fn):
            return {"ok": False, "failed_stage": name, "stages": st}
    return {"ok": True, "stages": st}


def mv_graph_selftest(model, batch, B):
    """Sharded multi-view job: the captured hipGraph of the step (with its collectives inside) must replay to what the eager step gives
    on the first step.  Under the gloo test hook the step is eager either way; the comparison then only exercises the code path."""
    unet = model.model.diffusion_model
    c_concat, c_cross, uc_cross, x_T = batch
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
    t = torch.full((2 * B,), 501, device=x.device, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half()
    with torch.no_grad():
        prev = unet.use_hip_graph
        unet.use_hip_graph = False
        try:
            eager = unet(x, t, ctx).float()
        finally:
            unet.use_hip_graph = prev
        g1 = unet(x, t, ctx).float()
        g2 = unet(x, t, ctx).float()
    torch.cuda.synchronize()
    d1, d2 = (g1 - eager).abs().max().item(), (g2 - g1).abs().max().item()
    fin = bool(torch.isfinite(g1).all())
    return {"ok": fin and d1 == 0.0 and d2 == 0.0, "max_abs_graph_vs_eager": d1, "max_abs_replay_vs_replay": d2, "finite": fin,
            "graph_captured": bool(prev and getattr(unet, "mv_shard_graph", False))}


def batch_sensitivity(model, h, w, device, Bs=(1, 2, 8), n=12, warm=3):
    """Extra key, not the metric: HIP-event time of ONE graph-replayed UNet forward at other per-GPU batches B (UNet batch
    2B under CFG), its whole-step fraction of the dense-fp16 MFMA peak, and how many GEMM shapes of that batch are not in the in-tree tile
    table (they take the library's static heuristic)."""
    from leftrefill_amd import ops
    from leftrefill_amd.flops import unet_flops
    unet = model.model.diffusion_model
    out = {}
    for B in Bs:
        batch = synthetic_batch(B, h, w, device, 4242 + B)
        c_concat, c_cross, uc_cross, x_T = batch
        x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
        t = torch.full((2 * B,), 501, device=device, dtype=torch.long)
        ctx = torch.cat([uc_cross, c_cross]).half()
        ops.TABLE_MISSES.clear()
        ops.TABLE_HITS.clear()
        evs = []
        with torch.no_grad():
            for i in range(warm + n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                unet(x, t, ctx)
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs[warm:])
        ms = ts[len(ts) // 2]
        fl = unet_flops(unet, x.shape[2], x.shape[3])
        tf = 2 * B * fl["total"] / (ms * 1e-3) / 1e12
        out[f"B{B}"] = {"unet_batch": 2 * B, "unet_forward_ms_median": ms, "ms_per_sample": ms / B, "tflops": tf, "frac_of_mfma_peak": tf / 2500.0,
                        "gemm_shapes": len(ops.TABLE_MISSES) + len(ops.TABLE_HITS), "gemm_shapes_not_in_tile_table": len(ops.TABLE_MISSES)}
    return out


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:      # noqa: BLE001  -- a multi-GPU failure must be diagnosable from the JSON line alone
        import traceback
        rank = int(os.environ.get("RANK", "0"))
        err = {"metric": "512x1024 stitched images/sec @ 50 DDIM steps, cfg=2.5; per-UNet-step ms", "value": None, "unit": "images/s",
               "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": {"stage": _STAGE[0], "rank": rank, "type": type(e).__name__,
                                                                         "detail": str(e)[:500], "traceback_tail": traceback.format_exc()[-1500:]}}
        if rank == 0 or _STAGE[0] in ("init_process_group", "selftest"):
            emit_json_line(err)
        print(f"[bench rank {rank}] failed in stage {_STAGE[0]}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.exit(1)


_STAGE = ["arguments"]


def _main():
    global S_DDIM
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="user batch B per GPU (UNet batch 2B under CFG)")
    ap.add_argument("--workload", default="single", choices=["single", "mv5", "mv4", "train"],
                    help="single = configs[1] (the metric); mv5 / mv4 = multi-view config 4 on one GPU; train = training step "
                         "(configs[4]-like) -- neither is the metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"],
                    help="train workload: 16-bit type of activations / packed weights (configs[4] names bf16; the metric "
                         "workloads are fp16 like the reference's autocast)")
    ap.add_argument("--train-graph", action="store_true", help="train workload: capture the whole step into one hipGraph")
    ap.add_argument("--recompute", action="store_true", help="train workload: recompute blocks in the backward (use_checkpoint)")
    ap.add_argument("--dump-kernels", default=None, help="write per-launch (shape, us, TFLOP/s) records as JSON lines")
    ap.add_argument("--mv-shard", action="store_true",
                    help="mv5 workload in its configs[3] form: ONE canvas per rank (--gpus 4 = view_num - 1), per-block RCCL "
                         "all-gather of the reference halves + broadcast of the target half, captured in the hipGraph")
    ap.add_argument("--split-cfg", action="store_true",
                    help="single workload with B < #GPUs: the unconditional / conditional UNet passes of the same samples on rank "
                         "pairs (2 j, 2 j + 1), one all-gather of the eps halves per DDIM step")
    ap.add_argument("--ddim-steps", type=int, default=S_DDIM, help="DDIM steps per sampling (the metric is quoted at 50)")
    ap.add_argument("--selftest", action="store_true",
                    help="multi-GPU first-contact checks before timing (run automatically when --gpus > 1; this flag also runs them, "
                         "trivially, at --gpus 1 and adds the graph-vs-eager comparison of the sharded multi-view step)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the batch-sensitivity side measurement (B = 1, 2, 8 per GPU)")
    ap.add_argument("--task", default="nvs", choices=["nvs", "refill"],
                    help="train workload: nvs = BASELINE configs[4] as stated (NVSLDM, prompt tokens + pose MLP); refill = RefInpaintLDM tokens only")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: re-launch as N ranks (one per GPU) under torch.distributed.run, like the driver does
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: one rank per GPU, the two must agree")
    S_DDIM = a.ddim_steps
    backend = None
    selftest = None
    _STAGE[0] = "init_process_group"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LR_BENCH_SHARE_GPU=1 is a test hook only: all ranks on cuda:0 over gloo, to exercise this code path on a 1-GPU box
        share = os.environ.get("LR_BENCH_SHARE_GPU") == "1"
        if share:
            local = 0
        torch.cuda.set_device(local)
        if share:
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == a.gpus
        backend = dist.get_backend()
        _STAGE[0] = "selftest"
        selftest = dist_selftest(rank, world, torch.device("cuda", torch.cuda.current_device()), backend, share)
        if not selftest["ok"]:      # every rank prints its own view: the collective that would gather them is what failed
            emit_json_line({"metric": "512x1024 stitched images/sec @ 50 DDIM steps, cfg=2.5; per-UNet-step ms", "value": None,
                            "unit": "images/s", "n_gpus": world,
                            "error": {"stage": "selftest:" + selftest["failed_stage"], "rank": rank, "selftest": selftest}})
            sys.exit(1)
    else:
        torch.cuda.set_device(0)
        if a.selftest:
            selftest = {"ok": True, "stages": {}, "note": "single GPU: no collectives to check"}
    device = torch.device("cuda", torch.cuda.current_device())
    _STAGE[0] = "workload"
    if a.workload == "train":
        _STAGE[0] = "train workload"
        res = train_bench(a, rank, world, device)
        if selftest is not None:
            res["selftest"] = selftest
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    # The metric times the FULL classifier-free-guidance batch: the sampler's exact shared-prefix optimisation (the context-free
    # first block computed once for the identical uncond / cond halves, UNetModel.cfg_shared_prefix, on by default for users)
    # is switched off for `value` and measured next to it as `cfg_shared_prefix` below.
    os.environ["LEFTREFILL_CFG_SHARED_PREFIX"] = "0"
    B, h, w = a.batch, 64, 128
    samples_per_step = B
    replicas = world            # independent copies of the workload (weak scaling): ranks / ranks-per-replica
    seed = 1234 + rank
    if a.workload != "single":      # B counts canvases from here on; one sample = (view_num - 1 | view_num) canvases
        mv = MV_WORKLOADS[a.workload]
        views = mv["view_num"] - 1 if mv["concat_target"] else mv["view_num"]
        samples_per_step = max(1, a.batch // 4)
        B, h, w = samples_per_step * views, mv["h"], mv["w"]
        if a.mv_shard:              # configs[3]: the canvases of a sample spread over the ranks, one each
            if a.workload != "mv5" or world not in (1, views):
                raise SystemExit(f"--mv-shard: --workload mv5 with --gpus {views} (one canvas [ref_i | target] per rank), or --gpus 1 for the "
                                 "per-rank cost with simulated peers")
            if world == 1:          # ONE rank of the 4-rank job on its own: the peers' rows are local copies (leftrefill_amd.dist._sim_world)
                os.environ["LEFTREFILL_MV_SIM_WORLD"] = str(views)
                os.environ.setdefault("LEFTREFILL_MV_SIM_RANK", "0")
            B, replicas = samples_per_step, 1
    if a.split_cfg:
        if a.workload != "single" or world % 2:
            raise SystemExit("--split-cfg: the single workload on an even number of ranks (pairs run uncond / cond)")
        replicas = world // 2
        seed = 1234 + rank // 2     # a pair works on the same samples and draws the same DDIM noise
        torch.manual_seed(4321 + rank // 2)
        torch.cuda.manual_seed(4321 + rank // 2)

    _STAGE[0] = "build_model"
    model = build_model(device, a.workload)
    batch = synthetic_batch(B, h, w, device, seed)
    if a.mv_shard:
        unet = model.model.diffusion_model
        unet.mv_shard, unet.mv_shard_graph = True, True     # the graph (incl. its RCCL collectives) is captured under `nccl` only
    if a.split_cfg:
        from leftrefill_amd import dist as lrd
        lrd.enable_split_cfg(True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if a.mv_shard and world > 1 and (a.selftest or backend == "nccl"):
        _STAGE[0] = "selftest:mv_graph_vs_eager"
        mvs = mv_graph_selftest(model, batch, B)
        flags = [None] * world
        torch.distributed.all_gather_object(flags, bool(mvs["ok"]))
        selftest["stages"]["mv_graph_vs_eager"] = dict(mvs, ranks_ok=flags)
        if not all(flags):
            selftest.update(ok=False, failed_stage="mv_graph_vs_eager")
            if rank == 0:
                emit_json_line({"metric": "multi-view samples/sec", "value": None, "unit": "samples/s", "n_gpus": world,
                                "error": {"stage": "selftest:mv_graph_vs_eager", "selftest": selftest}})
            sys.exit(1)
    # one-time preparation, like building the model: tile autotune + hipGraph capture for this shape (a 4-step sampling),
    # so that --warmup 0 does not put them inside the timed region
    _STAGE[0] = "prepare (first sampling: hipGraph capture)"
    sample_once(model, batch, B, steps=4)
    _STAGE[0] = "warmup"
    for _ in range(a.warmup):
        sample_once(model, batch, B)
    barrier()
    _STAGE[0] = "timed region"
    hw = HwSampler(device).start() if rank == 0 else None      # shader clock / socket power over exactly the timed region
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = sample_once(model, batch, B)
    barrier()
    dt = time.perf_counter() - t0
    hw_stats = hw.stop() if hw is not None else {}
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all()
    _STAGE[0] = "report"
    ms_per_step = 1e3 * dt / a.steps
    images_per_s = replicas * samples_per_step * a.steps / dt
    unet_step_ms = ms_per_step / S_DDIM     # per DDIM iteration (UNet step at batch 2B + fused update), incl. host loop

    res = {"metric": "512x1024 stitched images/sec @ 50 DDIM steps, cfg=2.5; per-UNet-step ms", "value": images_per_s,
           "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "configs[1]: 1-ref inpainting, 512x1024 canvas (latent 64x128), B=4 per GPU "
                                  "(UNet batch 8 under CFG), 50 DDIM steps, cfg=2.5, eta=1.0, fp16",
                      "global_batch": replicas * B, "per_gpu_batch": B, "ddim_steps": S_DDIM, "cfg": CFG, "eta": ETA,
                      "parallelism": f"dp{world} (sample-sharded, no data-path collective)",
                      "ranks": world, "backend": backend,
                      "note": "every DDIM step runs the full UNet at batch 2B (the exact shared-prefix optimisation of the sampler is OFF for this number); two loop-invariants are "
                              "computed once per sampling, inside the timed region, instead of once per step: the cross-attention K/V projection of the constant "
                              "context (3.9 of 1850 GFLOP per sample per forward, 0.2 %) and the timestep-embedding rows (time MLP + the 22 emb_layers: functions "
                              "of the schedule alone, 50 rows in 4 batched launches; LEFTREFILL_EMB_TABLE=0 recomputes them every step, +0.09 ms per step)"},
           "per_unet_step_ms": unet_step_ms}
    if selftest is not None:
        res["selftest"] = selftest
    if a.workload != "single":
        res["config"]["workload"] = (f"config 4 ({a.workload}): {MV_WORKLOADS[a.workload]}, {samples_per_step} sample(s) = {B} "
                                     f"canvases per GPU (UNet batch {2 * B}), {S_DDIM} DDIM steps, cfg=2.5, eta=1.0, fp16")
        res["config"]["global_batch"], res["config"]["per_gpu_batch"] = replicas * samples_per_step, samples_per_step
        if a.mv_shard and world == 1:
            views_ = MV_WORKLOADS[a.workload]["view_num"] - 1
            res["metric"] = "PER-RANK cost of the sharded multi-view step (rank %s of %d, peers simulated by local copies: no wire time)" % (
                os.environ.get("LEFTREFILL_MV_SIM_RANK", "0"), views_)
            res["unit"] = "samples/s if the collectives were free"
            from leftrefill_amd import engine as _eng
            res["config"]["parallelism"] = (f"mv-shard x{views_} simulated on one GPU: one canvas per rank; per transformer block the rows ONE "
                                            "all_gather_into_tensor of the ranks' canvases (+ LayerNorm statistics, same message) would deliver are "
                                            "written from local data (same kernels, same bytes); "
                                            + ("target query rows split over the ranks, their results all-gathered behind the out-projection "
                                               "(second collective, simulated the same way)" if _eng.MV_SPLIT_TARGET else
                                               "target rows replicated on every rank (LEFTREFILL_MV_SPLIT_TARGET=0)"))
            res["per_rank_unet_step_ms"] = unet_step_ms
        elif a.mv_shard:
            res["scaling"] = "strong"
            res["metric"] = "multi-view samples/sec (4-ref, 5 x 4096-token cross-view self-attention) @ 50 DDIM steps, cfg=2.5; canvases sharded over ranks"
            res["unit"] = "samples/s"
            res["config"]["parallelism"] = (f"mv-shard x{world}: one canvas per rank; per transformer block one all_gather_into_tensor of the "
                                            f"ranks' canvases with their LayerNorm statistics and one of the new target-row slices ({backend}), "
                                            + ("captured in the hipGraph" if backend == "nccl" else "eager (gloo test hook)"))
    if a.split_cfg:
        res["config"]["parallelism"] = (f"split-cfg x{world}: {replicas} pair(s) of ranks, uncond pass on rank 2j / cond pass on 2j+1 at UNet "
                                        f"batch {B}, one all-gather of the eps halves per DDIM step ({backend})")
        res["config"]["global_batch"] = replicas * B
    if S_DDIM != 50:
        res["config"]["note"] = f"NOT the metric's configuration: {S_DDIM} DDIM steps per sampling (test / smoke run)"

    if rank == 0 and not a.no_roofline and not (a.mv_shard and world == 1):      # (the instrumented eager step is not built for simulated peers)
        kern, fl = kernel_roofline(model, batch, B, a.dump_kernels)
        g = kern["gemm_conv"]
        traffic, traffic_note = None, "not measured (--no-traffic / multi-GPU / other workload)"
        if a.workload == "single" and world == 1 and not a.no_traffic:
            tr, err = measure_traffic(g["launches"])      # live: two rocprofv3 --pmc passes over one eager UNet step
            if tr is not None:
                traffic = tr["traffic_bytes_per_launch"]
                traffic_note = (f"measured in this run: FETCH_SIZE x2 {tr['fetch_bytes_per_launch'] / 1e6:.1f} MB + WRITE_SIZE "
                                f"{tr['write_bytes_per_launch'] / 1e6:.1f} MB per launch over {tr['fetch_launches']} launches")
            else:
                traffic_note = "measurement failed: " + str(err)
        res["roofline"] = {"bound": "mfma", "kernel": "gemm_conv_kernel<BN> / gemm_conv_pipe_kernel<BM,NW,BN,..> / conv_halo_kernel<BN,..> (implicit-GEMM conv3x3 / 1x1 / linear family incl. the halo-tile 3x3 conv, instance picked per shape by the tile table)",
                           "achieved": g["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": g["tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic,
                           "traffic_unit": "bytes/launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_note": traffic_note,
                           "algorithmic_bytes_per_launch": g["algorithmic_bytes_per_launch"],
                           "launches_per_unet_step": g["launches"], "avg_launch_us": g["avg_us"],
                           "algorithmic_gflop_per_unet_step": g["algorithmic_gflop"]}
        # what the chip gave during the timed region, and what its matrix cores sustain on random operands right now
        res["roofline"].update({k: hw_stats.get(k) for k in ("sclk_mhz_mean", "sclk_mhz_min", "sclk_mhz_max", "power_w_mean",
                                                              "power_w_max", "hw_samples", "hw_sampler")})
        try:
            pk = sustained_mfma_peak() if world == 1 else {"error": "single-GPU runs only"}
        except Exception as e:      # noqa: BLE001
            pk = {"error": f"{type(e).__name__}: {e}"[:300]}
        res["roofline"]["sustained_peak_measured"] = pk.get("tflops_random_16x16x32")
        res["roofline"]["sustained_peak_detail"] = pk
        if pk.get("tflops_random_16x16x32"):
            res["roofline"]["frac_of_sustained_peak"] = g["tflops"] / pk["tflops_random_16x16x32"]
        step_tflops = 2 * B * fl["total"] / (unet_step_ms * 1e-3) / 1e12
        res["kernel_table"] = kern.get("table", [])[:64]
        res["kernels"] = {"attention_kernel": kern["attention"], "xattn_block_kernel": kern.get("xattn_block"), "ffn_block_kernel": kern.get("ffn_block"), "stin_block_kernel": kern.get("stin_block"), "rowlin_kernel": kern.get("rowlin"),
                          "unet_step": {"algorithmic_tflop": 2 * B * fl["total"] / 1e12, "ms": unet_step_ms,
                                        "tflops": step_tflops, "frac_of_mfma_peak": step_tflops / MFMA_PEAK_TFLOPS}}
    if rank == 0 and not a.no_roofline and a.workload == "single":
        # side measurements, reported beside the metric and never inside `value`; a failure here must not cost the bench line
        def side(name, fn):
            try:
                res[name] = fn()
            except Exception as e:      # noqa: BLE001
                res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

        def shared_prefix():
            os.environ["LEFTREFILL_CFG_SHARED_PREFIX"] = "1"
            try:
                sample_once(model, batch, B, steps=4)        # capture the graph of the shared-prefix step
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n = max(1, min(a.steps, 3))
                for _ in range(n):
                    o = sample_once(model, batch, B)
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1) / n
                return {"images_per_s": B / dt1, "per_unet_step_ms": 1e3 * dt1 / S_DDIM, "finite": bool(torch.isfinite(o).all()),
                        "note": "same sampling with the context-free prefix of the UNet (conv_in, first ResBlock, first self-attention, "
                                "first cross-attention query projection) computed once for the identical uncond / cond halves -- bit-identical "
                                "results (tests/test_gpu_unet.py::test_cfg_shared_prefix_is_exact); not the metric"}
            finally:
                os.environ["LEFTREFILL_CFG_SHARED_PREFIX"] = "0"
        if a.workload == "single":
            side("cfg_shared_prefix", shared_prefix)
        side("unet_step_events", lambda: unet_step_events(model, batch, B))
        if world == 1 and not a.no_batch_sweep:
            side("batch_sensitivity", lambda: batch_sensitivity(model, h, w, device))
        side("vae_512x1024", lambda: vae_timing(B, device))
        v = res["vae_512x1024"]
        if "error" not in v:
            # caller view (log_images): 2 VAE encodes (image, masked image) + sampling + 1 decode per batch
            res["end_to_end_images_per_s_incl_vae"] = B / (ms_per_step + 2 * v["encode"]["ms"] + v["decode"]["ms"]) * 1e3
        if world == 1:      # next row 8f-2: one training step of the prompt tokens (see train_bench)
            def train():
                tr = train_bench(a, rank, world, device, model=model, steps=3)
                return {k: tr[k] for k in ("value", "unit", "ms_per_step", "forward_only_ms", "final_loss", "peak_memory_gib", "roofline")}
            side("training_256x512_b16", train)

            def train_bf16():
                # bf16 needs no loss-scale decisions on the host: the whole step (forward, HIP backward, AdamW) replays as ONE hipGraph
                # (the eager step is host-bound: ~1300 launches, 2.9 of 30.3 ms idle -- profiles/r05_train_breakdown.txt)
                a.dtype, prev_graph = "bf16", getattr(a, "train_graph", False)
                a.train_graph = True
                try:
                    tr = train_bench(a, rank, world, device, model=model, steps=3)
                    return {k: tr[k] for k in ("value", "unit", "ms_per_step", "forward_only_ms", "final_loss", "peak_memory_gib", "roofline", "hip_graph")}
                finally:
                    a.dtype, a.train_graph = "f16", prev_graph
            side("training_256x512_b16_bf16", train_bf16)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline()
        except Exception as e:      # noqa: BLE001
            res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
