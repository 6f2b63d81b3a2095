"""Generate tests/golden/*.npz by running the REAL reference on CPU (authoring container only).

    python -m oracle.make_golden [--only ops|unet|mv|sampler] [--skip-full]

Imports /root/reference through oracle/ref_import.py, fills it with the
name-keyed deterministic weights of oracle/weights.py and stores only the
reference *outputs* (inputs/weights are regenerated from their names).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from . import golden_spec as G
from . import ref_import, unet_ref, weights, ddim_ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(module, state):
    missing, unexpected = module.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    return module.eval()


def gen_ops(ns):
    A, O, U = ns.attention, ns.openaimodel, ns.dutil
    out = {}
    for name, kind, p in G.OP_CASES:
        st = G.op_state(name, kind, p)
        i = G.op_inputs(name, kind, p)
        with torch.no_grad():
            if kind == "gn_silu":
                mod = _load(U.normalization(p["C"]), st)
                y = torch.nn.functional.silu(mod(i["x"]))
            elif kind == "normalize":
                y = _load(A.Normalize(p["C"]), st)(i["x"])
            elif kind == "layernorm":
                y = _load(torch.nn.LayerNorm(p["C"]), st)(i["x"])
            elif kind == "conv3x3":
                y = _load(U.conv_nd(2, p["Cin"], p["Cout"], 3, padding=1), st)(i["x"])
            elif kind == "conv1x1":
                y = _load(U.conv_nd(2, p["Cin"], p["Cout"], 1), st)(i["x"])
            elif kind == "down":
                y = _load(O.Downsample(p["C"], True, dims=2, out_channels=p["C"]), st)(i["x"])
            elif kind == "up":
                y = _load(O.Upsample(p["C"], True, dims=2, out_channels=p["C"]), st)(i["x"])
            elif kind == "res":
                mod = _load(O.ResBlock(p["Cin"], 1280, 0, out_channels=p["Cout"], dims=2, use_checkpoint=True), st)
                y = mod(i["x"], i["emb"])
            elif kind == "attn":
                mod = _load(A.CrossAttention(p["C"], context_dim=p["ctx"], heads=p["heads"], dim_head=64), st)
                y = mod(i["x"], context=i.get("ctx"))
            elif kind == "ff":
                y = _load(A.FeedForward(p["C"], glu=True), st)(i["x"])
            elif kind == "tblock":
                mod = _load(A.BasicTransformerBlock(p["C"], p["heads"], 64, context_dim=p["ctx"]), st)
                y = mod(i["x"], context=i["ctx"])
            elif kind == "st":
                mod = _load(A.SpatialTransformer(p["C"], p["heads"], 64, depth=1, context_dim=p["ctx"],
                                                 use_linear=True), st)
                y = mod(i["x"], context=i["ctx"])
            else:
                raise ValueError(kind)
        out[name] = y.float().numpy()
        print(f"  op {name:18s} {tuple(y.shape)} mean {y.mean():+.4f} std {y.std():.4f}")
    # G1 timestep embedding
    t = torch.tensor([1, 21, 481, 981])
    out["timestep_embedding_320"] = U.timestep_embedding(t, 320).numpy()
    out["timestep_embedding_64"] = U.timestep_embedding(t, 64).numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)


def _taps_stats(taps):
    keys = sorted(taps.keys())
    return keys, np.array([[taps[k].mean().item(), taps[k].std().item(), taps[k].abs().max().item()] for k in keys],
                          dtype=np.float64)


def gen_unet(ns, skip_full=False):
    out = {}
    models = {}
    for case, cname, N, H, W, ts in G.UNET_CASES:
        if skip_full and cname == "FULL":
            continue
        cfg = G.CONFIGS[cname]
        if cname not in models:
            t0 = time.time()
            m = ns.openaimodel.UNetModel(**cfg.kwargs())
            _load(m, G.unet_state(cname))
            models[cname] = m
            print(f"  built {cname} in {time.time() - t0:.1f}s")
        m = models[cname]
        x, t, ctx = G.unet_inputs(case, cfg, N, H, W, ts)
        # forward hooks to record per-block outputs (reference naming: in{i} / mid / out{i})
        taps, hooks = {}, []
        for i, b in enumerate(m.input_blocks):
            hooks.append(b.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"in{i}", o.detach())))
        hooks.append(m.middle_block.register_forward_hook(lambda mod, a, o: taps.__setitem__("mid", o.detach())))
        for i, b in enumerate(m.output_blocks):
            hooks.append(b.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"out{i}", o.detach())))
        t0 = time.time()
        with torch.no_grad():
            y = m(x, t, ctx)
        for h in hooks:
            h.remove()
        keys, stats = _taps_stats(taps)
        out[case] = y.float().numpy()
        out[case + ".tap_keys"] = np.array(keys)
        out[case + ".tap_stats"] = stats
        print(f"  unet {case:22s} {tuple(y.shape)} mean {y.mean():+.4f} std {y.std():.4f} max {y.abs().max():.3f} "
              f"({time.time() - t0:.1f}s)")
    if skip_full:  # keep previously generated FULL entries
        p = os.path.join(OUT, "unet.npz")
        if os.path.exists(p):
            old = dict(np.load(p))
            old.update(out)
            out = old
    np.savez_compressed(os.path.join(OUT, "unet.npz"), **out)


def gen_mv(ns):
    out = {}
    for case, V, concat, b, H, W in G.MV_CASES:
        cfg = G.mv_config(V, concat)
        m = ns.multiview_unet.MultiViewUnetModel(**cfg.kwargs())
        shapes = unet_ref.param_shapes(cfg)
        _load(m, weights.fill_state_dict(shapes, prefix="unet.MV."))
        n = b * (V - 1 if concat else V)
        x, t, ctx = G.unet_inputs(case, cfg, n, H, W, [501] * n)
        with torch.no_grad():
            y = m(x, t, ctx)
        out[case] = y.float().numpy()
        print(f"  mv   {case:22s} {tuple(y.shape)} mean {y.mean():+.4f} std {y.std():.4f}")
    np.savez_compressed(os.path.join(OUT, "multiview.npz"), **out)


class _FakeLDM:
    """Just enough of LatentDiffusion for DDIMSampler (ddim.py:14,26-32,232,342)."""

    def __init__(self, ns, wrapper=None, fixed=None):
        ac = ddim_ref.alphas_cumprod()
        # reference: ddpm.py:149-169
        betas = np.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=np.float64) ** 2
        acp = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = 1000
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(acp, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, acp[:-1]), dtype=torch.float32)
        assert np.array_equal(self.alphas_cumprod.numpy(), ac)
        self.device = torch.device("cpu")
        self.parameterization = "eps"
        self.model = wrapper
        self.fixed = fixed
        self.ns = ns
        self.calls = []

    def apply_model(self, x, t, cond, **kw):
        self.calls.append(t.clone())
        if self.fixed is not None:
            return self.fixed
        return self.ns.ddpm.LatentDiffusion.apply_model(self, x, t, cond)


def gen_sampler(ns):
    out = {}
    # G2 schedule tables straight from the reference sampler
    for S in (10, 50):
        for eta in (0.0, 1.0):
            s = ns.ddim.DDIMSampler(_FakeLDM(ns))
            s.make_schedule(S, ddim_eta=eta, verbose=False)
            tag = f"sched_S{S}_eta{int(eta)}"
            out[tag + ".timesteps"] = np.asarray(s.ddim_timesteps)
            out[tag + ".alphas"] = np.asarray(s.ddim_alphas, dtype=np.float64)
            out[tag + ".alphas_prev"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
            out[tag + ".sigmas"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
            out[tag + ".sqrt_one_minus_alphas"] = np.asarray(s.ddim_sqrt_one_minus_alphas, dtype=np.float64)
    out["alphas_cumprod"] = _FakeLDM(ns).alphas_cumprod.numpy()
    # G6 single p_sample_ddim step with a fixed model output
    for case, S, eta, index in G.STEP_CASES:
        B, h, w = 2, 8, 16
        x = G.T(case + ".x", (B, 4, h, w))
        e = G.T(case + ".e", (2 * B, 4, h, w))
        noise = G.T(case + ".noise", (B, 4, h, w))
        ldm = _FakeLDM(ns, fixed=e)
        s = ns.ddim.DDIMSampler(ldm)
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        ns.ddim.noise_like = lambda shape, device, repeat=False: noise
        cond = {"c_concat": [torch.zeros(B, 5, h, w)], "c_crossattn": [torch.zeros(B, 77, 8)]}
        t = torch.full((B,), int(s.ddim_timesteps[index]), dtype=torch.long)
        x_prev, pred_x0 = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=G.CFG_SCALE,
                                          unconditional_conditioning=cond)
        out[case + ".x_prev"] = x_prev.numpy()
        out[case + ".pred_x0"] = pred_x0.numpy()
    # G7 trajectories through LatentDiffusion.apply_model + DiffusionWrapper('hybrid') + reference UNet (SMALL)
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    ucfg = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": cfg.kwargs()}
    wrapper = ns.ddpm.DiffusionWrapper(ucfg, "hybrid")
    _load(wrapper.diffusion_model, G.unet_state(G.TRAJ_CONFIG))
    for case, S, eta, B, h, w in G.TRAJ_CASES:
        x_T = G.T(case + ".x_T", (B, 4, h, w))
        c_concat = G.T(case + ".c_concat", (B, 5, h, w))
        c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim))
        uc_cross = G.T(case + ".uc_cross", (B, 77, cfg.context_dim))
        noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)) for i in range(S)]
        it = iter(noises)
        ns.ddim.noise_like = lambda shape, device, repeat=False: next(it)
        ldm = _FakeLDM(ns, wrapper=wrapper)
        s = ns.ddim.DDIMSampler(ldm)
        cond = {"c_concat": [c_concat], "c_crossattn": [c_cross]}
        uc = {"c_concat": [c_concat], "c_crossattn": [uc_cross]}
        samples, inter = s.sample(S, B, (4, h, w), cond, verbose=False, eta=eta, x_T=x_T,
                                  unconditional_guidance_scale=G.CFG_SCALE, unconditional_conditioning=uc)
        out[case + ".samples"] = samples.numpy()
        out[case + ".x_inter"] = torch.stack(inter["x_inter"]).numpy()
        out[case + ".pred_x0"] = torch.stack(inter["pred_x0"]).numpy()
        out[case + ".t_seq"] = torch.stack([c[0] for c in ldm.calls]).numpy()  # timestep fed to the UNet, per step
        assert all(c.shape[0] == 2 * B for c in ldm.calls)
        print(f"  traj {case:18s} samples mean {samples.mean():+.4f} std {samples.std():.4f} steps {len(ldm.calls)}")
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)


def gen_sampler_multi(ns):
    """`ddim_multi_sampling` (ddim.py:147-222): K conditionings sampled side by side, after every step the right half of
    ONE randomly picked condition (python `random.shuffle`) overwrites the right half of all of them."""
    import random
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    ucfg = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": cfg.kwargs()}
    wrapper = ns.ddpm.DiffusionWrapper(ucfg, "hybrid")
    _load(wrapper.diffusion_model, G.unet_state(G.TRAJ_CONFIG))
    out = {}
    for case, S, eta, B, h, w, K, seed in G.MULTI_CASES:
        x_T = [G.T(f"{case}.x_T{k}", (B, 4, h, w)) for k in range(K)]
        conds = [{"c_concat": [G.T(f"{case}.c_concat{k}", (B, 5, h, w))],
                  "c_crossattn": [G.T(f"{case}.c_cross{k}", (B, 77, cfg.context_dim))]} for k in range(K)]
        ucs = [{"c_concat": conds[k]["c_concat"], "c_crossattn": [G.T(f"{case}.uc_cross{k}", (B, 77, cfg.context_dim))]}
               for k in range(K)]
        noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)) for i in range(S * K)]
        it = iter(noises)
        ns.ddim.noise_like = lambda shape, device, repeat=False: next(it)
        ldm = _FakeLDM(ns, wrapper=wrapper)
        s = ns.ddim.DDIMSampler(ldm)
        random.seed(seed)
        samples, _ = s.sample(S, B, (4, h, w), conds, verbose=False, eta=eta, x_T=[t.clone() for t in x_T],
                              unconditional_guidance_scale=G.CFG_SCALE, unconditional_conditioning=ucs)
        out[case + ".samples"] = samples.numpy()
        out[case + ".t_seq"] = torch.stack([c[0] for c in ldm.calls]).numpy()
        assert len(ldm.calls) == S * K
        print(f"  multi {case}: samples mean {samples.mean():+.4f} std {samples.std():.4f} calls {len(ldm.calls)}")
    np.savez_compressed(os.path.join(OUT, "sampler_multi.npz"), **out)


def gen_train(ns):
    """Training objective + backward of the REAL reference on CPU (ddpm.py:900-935 p_losses through apply_model, the hybrid
    DiffusionWrapper and the reference UNet with its CheckpointFunction): loss, loss_dict and d loss / d context."""
    import types
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    ucfg = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": cfg.kwargs()}
    wrapper = ns.ddpm.DiffusionWrapper(ucfg, "hybrid")
    _load(wrapper.diffusion_model, G.unet_state(G.TRAJ_CONFIG))
    # NB the reference leaves the UNet parameters requiring grad (only the optimizer ignores them); its
    # CheckpointFunction.backward differentiates w.r.t. them and fails if they are frozen (util.py:133-151)
    out = {}
    for case, B, h, w, ts in G.TRAIN_CASES:
        fake = _FakeLDM(ns, wrapper=wrapper)
        fake.register_buffer = lambda name, val, persistent=True, _f=fake: setattr(_f, name, val)
        fake.v_posterior = 0.
        ns.ddpm.DDPM.register_schedule(fake, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120)
        fake.loss_type, fake.learn_logvar, fake.logvar = "l2", False, torch.zeros(1000)
        fake.l_simple_weight, fake.original_elbo_weight, fake.training = 1., 0., True
        fake.q_sample = types.MethodType(ns.ddpm.DDPM.q_sample, fake)
        fake.get_loss = types.MethodType(ns.ddpm.DDPM.get_loss, fake)
        x_start = G.T(case + ".x_start", (B, 4, h, w))
        noise = G.T(case + ".noise", (B, 4, h, w))
        c_concat = G.T(case + ".c_concat", (B, 5, h, w))
        c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim)).requires_grad_(True)
        t = torch.tensor(ts, dtype=torch.long)
        with torch.enable_grad():
            loss, ld = ns.ddpm.LatentDiffusion.p_losses(fake, x_start, {"c_concat": [c_concat], "c_crossattn": [c_cross]}, t,
                                                        noise=noise)
            loss.backward()
        out[case + ".loss"] = loss.detach().numpy()
        out[case + ".loss_simple"] = ld["train/loss_simple"].detach().numpy()
        out[case + ".loss_vlb"] = ld["train/loss_vlb"].detach().numpy()
        out[case + ".x_noisy"] = ns.ddpm.DDPM.q_sample(fake, x_start, t, noise).numpy()
        out[case + ".dctx"] = c_cross.grad.numpy()
        print(f"  train {case}: loss {float(loss):.6f} |dctx| max {c_cross.grad.abs().max():.3e}")
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)


VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def gen_vae(ns):
    """Host-side KL-VAE (reference ldm/models/autoencoder.py + diffusionmodules/model.py), reduced width."""
    from ldm.models.autoencoder import AutoencoderKL
    ref = AutoencoderKL(dict(VAE_DDCONFIG), {"target": "torch.nn.Identity"}, 4).eval()
    sd = {k: torch.from_numpy(weights.fill_like("vae." + k, v.shape)) for k, v in ref.state_dict().items()}
    ref.load_state_dict(sd)
    x = G.T("vae.x", (1, 3, 32, 64))
    with torch.no_grad():
        post = ref.encode(x)
        z = post.mode()
        np.savez_compressed(os.path.join(OUT, "vae.npz"), z=z.numpy(), dec=ref.decode(z).numpy(),
                            sample=post.sample().numpy(), keys=np.array(list(sd.keys())))


VAE_HIP_DDCONFIG = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
                        num_res_blocks=2, attn_resolutions=[], dropout=0.0)   # widths 64/128/256/256: multiples of 64


def gen_vae_hip(ns):
    """KL-VAE at channel widths the HIP kernels take (multiples of 64) + the two VAE-only operators at full width:
    AttnBlock(512) (model.py:153-204) and the asymmetric-pad Downsample(128) (model.py:69-88)."""
    from ldm.models.autoencoder import AutoencoderKL
    from ldm.modules.diffusionmodules import model as M
    ref = AutoencoderKL(dict(VAE_HIP_DDCONFIG), {"target": "torch.nn.Identity"}, 4).eval()
    sd = {k: torch.from_numpy(weights.fill_like("vaeh." + k, v.shape)) for k, v in ref.state_dict().items()}
    ref.load_state_dict(sd)
    x = G.T("vaeh.x", (2, 3, 64, 128))
    out = {}
    with torch.no_grad():
        moments = ref.quant_conv(ref.encoder(x))
        z = ref.encode(x).mode()
        out.update(moments=moments.numpy(), z=z.numpy(), dec=ref.decode(z).numpy())
        attn = M.AttnBlock(512).eval()
        attn.load_state_dict({k: torch.from_numpy(weights.fill_like("vaeh.attn." + k, v.shape))
                              for k, v in attn.state_dict().items()})
        out["attn_y"] = attn(G.T("vaeh.attn.x", (2, 512, 8, 16))).numpy()
        down = M.Downsample(128, True).eval()
        down.load_state_dict({k: torch.from_numpy(weights.fill_like("vaeh.down." + k, v.shape))
                              for k, v in down.state_dict().items()})
        out["down_y"] = down(G.T("vaeh.down.x", (2, 128, 16, 32))).numpy()
    np.savez_compressed(os.path.join(OUT, "vae_hip.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-full", action="store_true")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ns = ref_import.import_reference()
    todo = [a.only] if a.only else ["ops", "sampler", "sampler_multi", "train", "mv", "vae", "vae_hip", "unet"]
    for what in todo:
        print(f"[{what}]")
        t0 = time.time()
        {"ops": gen_ops, "unet": lambda n: gen_unet(n, a.skip_full), "mv": gen_mv, "sampler": gen_sampler,
         "sampler_multi": gen_sampler_multi, "train": gen_train, "vae": gen_vae, "vae_hip": gen_vae_hip}[what](ns)
        print(f"[{what}] done in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    sys.exit(main())
