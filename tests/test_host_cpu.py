"""CPU: C-ABI surface, host-side logic of the drop-in (schedules, packing, state-dict contract, sharding)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "leftrefill_hip.h")).read()
    src = re.sub(r"#ifdef LR_DEV_VARIANTS.*?#endif /\* LR_DEV_VARIANTS \*/", "", src, flags=re.S)      # developer builds only
    return sorted(set(re.findall(r"^(?:int|int64_t) (lr_\w+)\(", src, flags=re.M)))


def test_abi_library_exports_every_declared_symbol():
    from leftrefill_amd import _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/leftrefill_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    assert _lib.load().lr_abi_version() == _lib.ABI_VERSION


def test_product_library_has_no_switches():
    """SURVEY section 8b "no global state inside": nothing in csrc/ reads the environment, and the product build exports neither the
    developer knob table nor the measured-and-lost variants (they exist under -DLR_DEV_VARIANTS only, tools/build_variant.sh)."""
    from leftrefill_amd import _lib, build
    build.build(verbose=False)
    csrc = os.path.join(ROOT, "leftrefill_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _lib.DEV_SIGNATURES:
        assert not hasattr(lib, s), f"{s} must not be exported by the product library"


def test_gemm_args_struct_layout_matches_header():
    """sizeof/offsets of lr_gemm_args as the C compiler lays it out vs the ctypes mirror."""
    import subprocess, tempfile
    from leftrefill_amd._lib import GemmArgs
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "leftrefill_hip.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(lr_gemm_args), offsetof(lr_gemm_args, wt),
  offsetof(lr_gemm_args, bias), offsetof(lr_gemm_args, resid), offsetof(lr_gemm_args, out), offsetof(lr_gemm_args, tile_n)); }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")],
                       check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    got = [ctypes.sizeof(GemmArgs), GemmArgs.wt.offset, GemmArgs.bias.offset, GemmArgs.resid.offset,
           GemmArgs.out.offset, GemmArgs.tile_n.offset]
    assert [int(v) for v in out] == got


def test_attn_bwd_args_struct_layout_matches_header():
    import subprocess, tempfile
    from leftrefill_amd._lib import AttnBwdArgs
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "leftrefill_hip.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(lr_attn_bwd_args), offsetof(lr_attn_bwd_args, qt),
  offsetof(lr_attn_bwd_args, dq), offsetof(lr_attn_bwd_args, ldq), offsetof(lr_attn_bwd_args, B), offsetof(lr_attn_bwd_args, scale)); }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")],
                       check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    got = [ctypes.sizeof(AttnBwdArgs), AttnBwdArgs.qt.offset, AttnBwdArgs.dq.offset, AttnBwdArgs.ldq.offset,
           AttnBwdArgs.B.offset, AttnBwdArgs.scale.offset]
    assert [int(v) for v in out] == got


def test_xattn_args_struct_layout_matches_header():
    import subprocess, tempfile
    from leftrefill_amd._lib import XattnArgs
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "leftrefill_hip.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(lr_xattn_args), offsetof(lr_xattn_args, k), offsetof(lr_xattn_args, ldk),
  offsetof(lr_xattn_args, vt), offsetof(lr_xattn_args, stats_out), offsetof(lr_xattn_args, M), offsetof(lr_xattn_args, scale)); }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")],
                       check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    got = [ctypes.sizeof(XattnArgs), XattnArgs.k.offset, XattnArgs.ldk.offset, XattnArgs.vt.offset, XattnArgs.stats_out.offset,
           XattnArgs.M.offset, XattnArgs.scale.offset]
    assert [int(v) for v in out] == got


def test_xattn_k_slot_permutation():
    """packing.xattn_perm is the accumulator -> B-operand hand-off order of a 16x16x32 MFMA tile pair (csrc/xattn_block.hip)."""
    import torch
    from leftrefill_amd import packing
    perm = packing.xattn_perm(128)
    assert sorted(perm.tolist()) == list(range(128))
    # lane group f of k-step p holds, in slots 0..7, the four channels 4 f .. 4 f + 3 of tile 2p and of tile 2p + 1
    for h in range(2):
        for p in range(2):
            for f in range(4):
                got = perm[h * 64 + 32 * p + 8 * f: h * 64 + 32 * p + 8 * f + 8].tolist()
                want = [h * 64 + 16 * (2 * p) + 4 * f + r for r in range(4)] + [h * 64 + 16 * (2 * p + 1) + 4 * f + r for r in range(4)]
                assert got == want
    wk, wo = torch.arange(128.0)[:, None].repeat(1, 8), torch.arange(128.0)[None, :].repeat(4, 1)
    xk, xwo = packing.pack_xattn(wk, wo, torch.float32)
    assert torch.equal(xk[:, 0], perm.float())
    assert xwo.shape == (2, 4, 64) and torch.equal(xwo[0, 0], perm[:64].float()) and torch.equal(xwo[1, 3], perm[64:].float())


def test_gemm_plan_is_a_pure_function_of_the_shape():
    """Tile / split-K / pipeline selection never depends on timing: the committed table (tile_table.json) answers the shapes
    of the shipped workloads, the library's static heuristic (lr_gemm_plan, host code -- no GPU needed) everything else;
    `plan_batch_scale` plans a half batch exactly like the full one (the shared CFG prefix relies on it)."""
    import ctypes
    import json
    from leftrefill_amd import _lib, ops
    assert not ops.AUTOTUNE
    table = json.load(open(ops.TILE_TABLE_PATH))
    assert len(table) >= 200
    valid = {(128, 64), (128, 128), (128, 160), (256, 128), (256, 160), (256, 256), (256, 320)}
    for key, plan in table.items():
        assert len(key.split(",")) == 12 and (plan[0], plan[1]) in valid and plan[2] >= 1, (key, plan)
        pipe = plan[3] if len(plan) > 3 else 0
        # 4 = the 8-wave 4-stage 128-row instance; 8 = the halo-tile conv (3x3, stride 1, no upsample; 256 x {160, 320})
        assert pipe in (0, 4, 8)
        assert pipe != 4 or (plan[0] == 128 and plan[1] in (128, 160))
        assert pipe != 8 or (plan[0] == 256 and plan[1] in (160, 320) and key.split(",")[3:6] == ["9", "1", "0"])
        M, N, K, taps, stride, up, geglu, cat, asym, gelu, ln, stats = map(int, key.split(","))
        got = ops.gemm_plan(M, N, K, taps=taps, stride=stride, up=up, geglu=bool(geglu), concat=bool(cat), asym=bool(asym),
                            gelu=bool(gelu), ln=bool(ln), stats=bool(stats))
        assert tuple(got[:3]) == tuple(plan[:3])
    # untabulated shapes: the heuristic, twice the same answer, and a sane one
    lib = _lib.load()
    for M, N, K in ((777, 192, 1088), (3000, 320, 2880), (50000, 640, 5760), (96, 1280, 23040)):
        p1, p2 = ops.gemm_plan(M, N, K), ops.gemm_plan(M, N, K)
        assert p1 == p2 and (p1[0], p1[1]) in valid and 1 <= p1[2] <= 8, (M, N, K, p1)
    # the plan of a scaled batch is the plan of the larger shape
    a = _lib.GemmArgs()
    a.B, a.H, a.W, a.N, a.taps, a.C1 = 4, 64, 128, 320, 9, 320
    small, big = (ctypes.c_int32 * 4)(), (ctypes.c_int32 * 4)()
    lib.lr_gemm_plan(a, small)
    a.B = 8
    lib.lr_gemm_plan(a, big)
    with ops.plan_batch_scale(2):
        assert ops._PLAN_BATCH_SCALE[0] == 2
    assert ops._PLAN_BATCH_SCALE[0] == 1 and tuple(big)[:2] in valid and tuple(small)[:2] in valid


def test_missing_library_fails_loudly(monkeypatch):
    from leftrefill_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libleftrefill_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_ops_reject_cpu_tensors():
    from leftrefill_amd import ops
    with pytest.raises(AssertionError):
        ops.layer_norm(torch.zeros(4, 64, dtype=torch.float16), torch.ones(64), torch.zeros(64))


def test_dropin_schedule_tables_bit_exact(golden):
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import DDPM
    g = golden("sampler")
    m = DDPM({"target": "torch.nn.Identity"}, conditioning_key="hybrid", linear_start=0.00085, linear_end=0.0120)
    assert np.array_equal(m.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert m.num_timesteps == 1000 and m.parameterization == "eps"
    for S in (10, 50):
        for eta in (0.0, 1.0):
            s = DDIMSampler(m)
            s.make_schedule(S, ddim_eta=eta, verbose=False)
            tag = f"sched_S{S}_eta{int(eta)}"
            assert np.array_equal(s.ddim_timesteps, g[tag + ".timesteps"])
            for k, a in (("alphas", s.ddim_alphas), ("alphas_prev", s.ddim_alphas_prev), ("sigmas", s.ddim_sigmas),
                         ("sqrt_one_minus_alphas", s.ddim_sqrt_one_minus_alphas)):
                assert np.array_equal(np.asarray(a, dtype=np.float64), g[f"{tag}.{k}"]), (tag, k)


def test_dropin_timestep_embedding_host_branch(golden):
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.util import timestep_embedding
    out = timestep_embedding(torch.tensor([1, 21, 481, 981]), 320).numpy()
    np.testing.assert_allclose(out, golden("ops")["timestep_embedding_320"], rtol=0, atol=2e-6)


def test_state_dict_contract_full_config():
    """686 tensors, reference names and shapes -- built on the meta device."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    with torch.device("meta"):
        m = UNetModel(**unet_ref.FULL.kwargs())
    sd = m.state_dict()
    shapes = unet_ref.param_shapes(unet_ref.FULL)
    assert len(sd) == 686 and set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sd["input_blocks.0.0.weight"].shape == (320, 9, 3, 3)
    assert sd["input_blocks.4.0.skip_connection.weight"].shape == (640, 320, 1, 1)
    assert sum(v.numel() for v in sd.values()) == 865_925_124
    from leftrefill_amd.flops import unet_flops
    f = unet_flops(m, 64, 128)
    assert abs(f["total"] / 1e9 - 1849.75) < 0.01 and abs(f["attn"] / 1e9 - 497.09) < 0.01


def test_pack_conv_layout():
    from leftrefill_amd import packing
    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, 5, 3, 3, generator=g)
    x = torch.randn(2, 5, 6, 7, generator=g)
    wp = packing.pack_conv(w, cin_pad=8, cout_pad=8).float()          # [8, 9*8], k = tap*8 + c
    cols = F.unfold(F.pad(x, (0, 0, 0, 0, 0, 3)), 3, padding=1)        # [2, 8*9, 42], row = c*9 + tap
    cols = cols.reshape(2, 8, 9, 42).permute(0, 2, 1, 3).reshape(2, 72, 42)   # -> tap*8 + c
    y = torch.einsum("nk,bkp->bnp", wp, cols.half().float()).reshape(2, 8, 6, 7)
    ref = F.conv2d(x.half().float(), w.half().float(), padding=1)
    assert torch.allclose(y, ref, atol=1e-4)


def test_geglu_permutation():
    from leftrefill_amd import packing
    perm = packing.geglu_perm(64)
    assert sorted(perm.tolist()) == list(range(128))
    assert perm[:16].tolist() == list(range(16)) and perm[16:32].tolist() == list(range(64, 80))
    assert perm[32:48].tolist() == list(range(16, 32))


def test_dropin_rejects_unsupported_variants():
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import ResBlock, UNetModel
    with pytest.raises(NotImplementedError):
        ResBlock(64, 256, 0, use_scale_shift_norm=True)
    kw = unet_ref.MID.kwargs()
    kw["num_classes"] = 10
    with pytest.raises(NotImplementedError):
        UNetModel(**kw)


def test_create_model_from_yaml(tmp_path):
    """inpainting_ldm.model.create_model on a YAML with dotted-path targets (the plugin mechanism)."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.model import create_model
    cfg = tmp_path / "m.yaml"
    cfg.write_text("""
model:
  target: inpainting_ldm.ref_inpainting_ldm.RefInpaintLDM
  params:
    linear_start: 0.00085
    linear_end: 0.0120
    timesteps: 1000
    first_stage_key: image
    cond_stage_key: txt
    channels: 4
    conditioning_key: hybrid
    scale_factor: 0.18215
    use_ema: False
    unet_config:
      target: ldm.modules.diffusionmodules.openaimodel.UNetModel
      params: {use_checkpoint: True, image_size: 32, in_channels: 9, out_channels: 4, model_channels: 64,
               attention_resolutions: [2, 1], num_res_blocks: 1, channel_mult: [1, 2], num_head_channels: 64,
               use_spatial_transformer: True, use_linear_in_transformer: True, transformer_depth: 1,
               context_dim: 128, legacy: False}
    first_stage_config: {target: torch.nn.Identity}
    cond_stage_config: {target: torch.nn.Identity, params: {}}
    data_config: {img_size: 512, cfg: 2.5}
    save_prompt_only: True
""")
    m = create_model(str(cfg))
    assert type(m).__name__ == "RefInpaintLDM" and m.save_prompt_only and m.img_size == 512
    assert m.concat_keys == ("mask", "masked_image") and m.masked_image_key == "masked_image" and m.channels == 4
    assert any(k.startswith("model.diffusion_model.input_blocks.0.0") for k in m.state_dict())


# ---- world_size-2 gloo test of the sample sharding ---------------------------------------------------------------
def _worker(rank, world, port, q):
    import torch.distributed as dist
    from leftrefill_amd import dist as lrd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 5   # ragged: 3 + 2
    cond = {"c_concat": [torch.arange(B * 2.).reshape(B, 2)], "c_crossattn": [torch.arange(B * 3.).reshape(B, 3) + 100]}
    x_T = torch.arange(B * 4.).reshape(B, 4)

    def fake_sampler(c, uc, xt, b):
        assert c["c_concat"][0].shape[0] == b == xt.shape[0] == uc["c_crossattn"][0].shape[0]
        return xt * 2 + c["c_concat"][0].sum(1, keepdim=True) + c["c_crossattn"][0].sum(1, keepdim=True)

    out = lrd.sample_sharded(fake_sampler, cond, cond, x_T, B)
    full = fake_sampler(cond, cond, x_T, B)
    q.put((rank, bool(torch.equal(out, full)), lrd.shard_range(B, rank, world)))
    dist.destroy_process_group()


def test_sample_sharding_gloo_world2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res[0] == (0, True, (0, 3)) and res[1] == (1, True, (3, 5))


def _split_cfg_worker(rank, world, port, q):
    import torch.distributed as dist
    from leftrefill_amd import dist as lrd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lrd.enable_split_cfg(True)
    role = lrd.split_cfg_role()
    e_local = torch.full((3, 4, 2, 2), float(10 * (rank // 2) + role))       # eps half of this rank's pass
    eps = lrd.cfg_exchange(e_local)
    ok = (eps.shape == (6, 4, 2, 2) and bool(torch.all(eps[:3] == 10.0 * (rank // 2))) and bool(torch.all(eps[3:] == 10.0 * (rank // 2) + 1)))
    q.put((rank, role, ok, lrd.split_cfg_active()))
    lrd.enable_split_cfg(False)
    dist.destroy_process_group()


def test_split_cfg_pairs_gloo_world4():
    """cond / uncond passes on rank pairs (2 j, 2 j + 1): roles, pair-local all-gather with the unconditional half first."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_cfg_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(4))
    for p in procs:
        p.join(60)
    assert res == [(0, 0, True, True), (1, 1, True, True), (2, 0, True, True), (3, 1, True, True)]


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    from leftrefill_amd import dist as lrd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = torch.nn.Parameter(torch.zeros(73, 8))
    b = torch.nn.Parameter(torch.zeros(5))
    c = torch.nn.Parameter(torch.zeros(3))          # no gradient: skipped
    a.grad = torch.full((73, 8), float(rank + 1))
    b.grad = torch.arange(5.) * (rank + 1)
    lrd.allreduce_mean_grads([a, b, c])
    q.put((rank, bool(torch.all(a.grad == 1.5)), bool(torch.equal(b.grad, torch.arange(5.) * 1.5)), c.grad is None))
    dist.destroy_process_group()


def test_training_gradient_allreduce_gloo_world2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, True, True), (1, True, True, True)]


def test_eval_glue_paste_crop_psnr():
    """test_inpainting.py:143-158: paste known pixels, keep the target half, area down-sampling, per-image PSNR."""
    from leftrefill_amd import evalglue
    g = torch.Generator().manual_seed(0)
    N, H, W = 2, 16, 32
    out = {"pred": torch.rand(N, 3, H, W, generator=g) * 2 - 1, "origin_image": torch.rand(N, 3, H, W, generator=g) * 2 - 1}
    mask = (torch.rand(N, H, W, 1, generator=g) < 0.4).float()
    pred, origin = evalglue.compose_prediction(out, mask, test_size=16, metric_size=8)
    m = mask.permute(0, 3, 1, 2)
    full = out["pred"] * m + out["origin_image"] * (1 - m)
    assert torch.allclose(full[..., W // 2:].reshape(N, 3, 8, 2, 8, 2).mean((3, 5)), pred, atol=1e-6)   # area == 2x2 mean
    assert torch.allclose(out["origin_image"][..., W // 2:].reshape(N, 3, 8, 2, 8, 2).mean((3, 5)), origin, atol=1e-6)
    assert torch.equal((full == out["origin_image"]) | (m > 0).expand_as(full), torch.ones_like(full, dtype=torch.bool))
    same, _ = evalglue.compose_prediction(out, mask, test_size=16, metric_size=16)
    assert same.shape == (N, 3, 16, 16) and torch.equal(same, full[..., W // 2:])
    ps = evalglue.psnr01(pred, origin)
    ref = [10 * np.log10(1.0 / (((pred[i] - origin[i]) / 2) ** 2).mean().item()) for i in range(N)]
    np.testing.assert_allclose(ps.numpy(), ref, rtol=1e-5)
    sq, _ = evalglue.compose_prediction({"pred": out["pred"][..., :H], "origin_image": out["origin_image"][..., :H]},
                                        mask[:, :, :H])
    assert sq.shape == (N, 3, H, H)          # square inputs are not cropped


def test_shard_range_partitions():
    from leftrefill_amd.dist import shard_range
    for total in (1, 4, 7, 32):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_dropin_vae_matches_reference_golden(golden):
    """Host-side KL-VAE drop-in (PyTorch code, off the hot path): same 248-key naming scheme and outputs as the reference."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.models.autoencoder import AutoencoderKL
    from oracle import golden_spec as G, weights
    from oracle.make_golden import VAE_DDCONFIG
    g = golden("vae")
    m = AutoencoderKL(dict(VAE_DDCONFIG), {"target": "torch.nn.Identity"}, 4).eval()
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    m.load_state_dict({k: torch.from_numpy(weights.fill_like("vae." + k, v.shape)) for k, v in m.state_dict().items()})
    x = G.T("vae.x", (1, 3, 32, 64))
    with torch.no_grad():
        post = m.encode(x)
        z = post.mode()
        np.testing.assert_allclose(z.numpy(), g["z"], atol=1e-5)
        np.testing.assert_allclose(m.decode(z).numpy(), g["dec"], atol=5e-5)
        np.testing.assert_allclose(post.sample().numpy(), g["sample"], atol=1e-5)   # RNG re-seeded to 42 inside


def test_dropin_vae_matches_reference_golden_hip_widths(golden):
    """Same pin at the channel widths the HIP path takes (64/128/256/256, incl. the mid AttnBlock): the PyTorch definition
    that tests/test_gpu_vae.py uses as its second checker."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.models.autoencoder import AutoencoderKL
    from ldm.modules.diffusionmodules.model import AttnBlock, Downsample
    from oracle import golden_spec as G, weights
    from oracle.make_golden import VAE_HIP_DDCONFIG
    g = golden("vae_hip")

    def fill(mod, prefix):
        mod.load_state_dict({k: torch.from_numpy(weights.fill_like(prefix + k, v.shape))
                             for k, v in mod.state_dict().items()})
        return mod.eval()

    m = fill(AutoencoderKL(dict(VAE_HIP_DDCONFIG), {"target": "torch.nn.Identity"}, 4), "vaeh.")
    x = G.T("vaeh.x", (2, 3, 64, 128))
    with torch.no_grad():
        post = m.encode(x)
        np.testing.assert_allclose(torch.cat([post.mean, post.logvar], 1).numpy(), g["moments"], atol=2e-5)
        np.testing.assert_allclose(m.decode(torch.from_numpy(g["z"])).numpy(), g["dec"], atol=5e-5)
        np.testing.assert_allclose(fill(AttnBlock(512), "vaeh.attn.")(G.T("vaeh.attn.x", (2, 512, 8, 16))).numpy(),
                                   g["attn_y"], atol=2e-5)
        np.testing.assert_allclose(fill(Downsample(128, True), "vaeh.down.")(G.T("vaeh.down.x", (2, 128, 16, 32))).numpy(),
                                   g["down_y"], atol=2e-5)


def test_full_vae_key_count():
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)   # model_config.yaml:44-58
    with torch.device("meta"):
        m = AutoencoderKL(dd, {"target": "torch.nn.Identity"}, 4)
    sd = m.state_dict()
    assert len(sd) == 248 and sd["encoder.conv_out.weight"].shape == (8, 512, 3, 3)
    assert sd["decoder.up.3.upsample.conv.weight"].shape == (512, 512, 3, 3)
    assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 83.65) < 0.01


# ---- canvas-sharded multi-view attention: exchange / row-ownership / write-back logic on a world_size-2 gloo group ----
def _mv_worker(rank, world, port, q):
    import torch.distributed as dist
    from leftrefill_amd import dist as lrd
    from oracle import golden_spec as G, unet_ref, weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, v, s, C, heads = 2, world, 4, 64, 2
    L = 2 * s * s
    x_full = G.T("mvd.x", (b * v, L, C))                       # '(b v) hw c' batch of canvases
    sd = weights.fill_state_dict({"m.attn1.to_q.weight": (C, C), "m.attn1.to_k.weight": (C, C),
                                  "m.attn1.to_v.weight": (C, C), "m.attn1.to_out.0.weight": (C, C),
                                  "m.attn1.to_out.0.bias": (C,), "m.norm1.weight": (C,), "m.norm1.bias": (C,)},
                                 prefix="mvd.")
    mode = unet_ref._Mode("fp32")
    # single-process oracle: gather -> LN -> attn1 + residual -> scatter (multiview_attention.py:436-462)
    seq, info = unet_ref.mv_gather(x_full, v + 1, True, False)
    n1 = unet_ref.layer_norm(seq, sd["m.norm1.weight"], sd["m.norm1.bias"])
    ref = unet_ref.mv_scatter(unet_ref.cross_attention(sd, "m.attn1", n1, n1, heads, mode) + seq, v + 1, True, False, info)
    ref = ref.reshape(b, v, L, C)[:, rank]
    # sharded: this rank only holds canvas `rank` of every sample
    x_local = x_full.reshape(b, v, L, C)[:, rank].contiguous()
    x_all = lrd.mv_all_gather_canvases(x_local)
    assert torch.equal(lrd.mv_sequence_from_canvases(x_all, s), seq)          # whole-canvas exchange (reference of the helper)
    sq = lrd.mv_gather_sequence(x_local, s)                                    # minimal exchange: ref halves + rank 0's target
    assert torch.equal(sq, seq)
    one = lrd.mv_gather_sequence(x_local[:1].contiguous(), s)                  # b == 1: receives straight into the sequence buffer
    assert torch.equal(one, seq[:1])
    buf = torch.empty(b, (v + 1) * s * s, C)
    assert lrd.mv_gather_sequence(x_local, s, seq=buf).data_ptr() == buf.data_ptr() and torch.equal(buf, seq)
    n = unet_ref.layer_norm(sq, sd["m.norm1.weight"], sd["m.norm1.bias"])
    qq = torch.nn.functional.linear(n, sd["m.attn1.to_q.weight"])
    kk = torch.nn.functional.linear(n, sd["m.attn1.to_k.weight"])
    vv = torch.nn.functional.linear(n, sd["m.attn1.to_v.weight"])
    a = unet_ref.attention(lrd.mv_own_rows(qq, rank, s), kk, vv, heads, mode)
    y = torch.nn.functional.linear(a, sd["m.attn1.to_out.0.weight"], sd["m.attn1.to_out.0.bias"]) + lrd.mv_own_rows(sq, rank, s)
    out = lrd.mv_canvas_from_own(y, s)
    err = float((out - ref).abs().max())
    # round 5: ONE collective for the inputs (whole canvases + a per-row fp32 payload packed into the same message) ...
    extra = G.T("mvd.extra", (b * v, L, 6)).reshape(b, v, L, 6)
    xa, ea = lrd.mv_exchange_canvases(x_local.half(), extra[:, rank].contiguous())
    assert xa.shape == (b, v, L, C) and torch.equal(xa, x_full.reshape(b, v, L, C).half()) and torch.equal(ea, extra)
    assert torch.equal(lrd.mv_sequence_from_canvases(xa, s), seq.half())
    assert lrd.mv_exchange_canvases(x_local.half())[1] is None
    # ... and the target query rows split over the ranks, the new target slices all-gathered behind the out-projection
    n_t = s * s // world
    a2 = unet_ref.attention(lrd.mv_own_rows_split(qq, rank, s, world), kk, vv, heads, mode)
    y2 = torch.nn.functional.linear(a2, sd["m.attn1.to_out.0.weight"], sd["m.attn1.to_out.0.bias"]) + lrd.mv_own_rows_split(sq, rank, s, world)
    tgt = lrd.mv_gather_target(y2[:, :n_t].contiguous())
    assert tgt.shape == (b, s * s, C)
    out2 = lrd.mv_canvas_from_own(torch.cat([tgt, y2[:, n_t:]], dim=1), s)
    q.put((rank, max(err, float((out2 - ref).abs().max()))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multiview_canvas_sharding_gloo(world):
    """Exchange / row-ownership / write-back logic of the canvas-sharded multi-view attention on world_size-2 and -4 gloo groups:
    the minimal exchange of round 3 (reference halves + rank 0's target), the one-message exchange of round 5 (whole canvases with
    the LayerNorm statistics packed behind each row) and the split of the target rows with its all-gather of the results."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert [r for r, _ in res] == list(range(world))
    assert all(err < 1e-5 for _, err in res), res


# ---- index tables of the sharded multi-view block against the torch glue, with DISTINCT canvases per rank ----------
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("split", [True, False])
def test_mv_shard_plan_tables_match_torch_glue_with_distinct_rank_canvases(world, split):
    """The row-copy tables of leftrefill_amd.dist.mv_shard_plan (seq_src, own_src, ref / tgt write-back) applied with index_select to
    a canvas-major message whose canvases all differ, against mv_sequence_from_canvases / mv_own_rows[_split] / mv_gather_target's
    layout / mv_canvas_from_own -- the simulated-peer GPU test delivers copies of the local rows and could not see a wrong canvas or
    rank index (reference semantics: ldm/modules/multiview_attention.py:436-462)."""
    from leftrefill_amd import dist as lrd
    v, b, s, C = world, 2, 4, 3
    s2, T = s * s, 2 * s * s
    g = torch.Generator().manual_seed(100 * world + int(split))
    canv = torch.randn(v, b, T, C, generator=g)               # canvas j of local sample bi, as rank j sends it
    msg = canv.reshape(v * b * T, C)                          # canvas-major receive buffer [v][b][T]
    x_all = canv.transpose(0, 1).contiguous()                 # [b, v, T, C]
    seq_ref = lrd.mv_sequence_from_canvases(x_all, s)
    for rank in range(world):
        p = lrd.mv_shard_plan(b, v, s, rank, split, "cpu")
        assert torch.equal(msg[p["seq_src"].long()], seq_ref.reshape(-1, C))
        own_ref = lrd.mv_own_rows_split(seq_ref, rank, s, v) if split else lrd.mv_own_rows(seq_ref, rank, s)
        assert p["Lo"] == own_ref.shape[1]
        assert torch.equal(msg[p["own_src"].long()], own_ref.reshape(-1, C))
        if not split:
            continue
        n = p["n"]
        y = torch.randn(b, p["Lo"], C, generator=g)           # this rank's new rows [target slice `rank`, ref_rank]
        yts = torch.randn(v, b, n, C, generator=g)            # the all-gathered target slices [v][b][n] (rank-major message)
        yts[rank] = y[:, :n]
        tgt_full = yts.permute(1, 0, 2, 3).reshape(b, v * n, C)      # what mv_gather_target returns
        canvas_ref = lrd.mv_canvas_from_own(torch.cat([tgt_full, y[:, n:]], dim=1), s)
        out = torch.full((b * T, C), float("nan"))
        out[p["ref_dst"].long()] = y.reshape(-1, C)[p["ref_src"].long()]
        out[p["tgt_dst"].long()] = yts.reshape(-1, C)[p["tgt_src"].long()]
        assert torch.equal(out, canvas_ref.reshape(-1, C)), (world, rank)


# ---- bench.py's multi-GPU first-contact checks, run over gloo on CPU ranks -------------------------------------------------------
def _selftest_worker(rank, world, port, q, fail):
    import importlib.util
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if fail:
        os.environ["LR_BENCH_SELFTEST_FAIL"] = fail
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("lr_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = bench.dist_selftest(rank, world, torch.device("cpu"), "gloo", share=False)
    q.put((rank, st))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [None, "all_reduce_sum"])
def test_bench_dist_selftest_gloo_world2(fail):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selftest_worker, args=(r, 2, port, q, fail)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    for r in range(2):
        st = res[r]
        if fail is None:
            assert st["ok"] and st["stages"]["all_gather_rank_stamp"]["ranks_seen"] == [0, 1] and st["stages"]["device_uniqueness"]["ok"]
        else:
            assert not st["ok"] and st["failed_stage"] == fail and "forced failure" in st["stages"][fail]["error"]
            assert st["stages"]["all_gather_into_tensor_f16"]["ok"]


def test_split_cols_backward_is_one_concatenation():
    """train_ops.split_cols (the per-block slices of the batched context K|V projection): column-slice views forward, the gradient of the wide
    tensor is the concatenation of the slices' gradients; an unused slice contributes zeros.  Pure torch: runs on CPU."""
    import torch
    from leftrefill_amd import train_ops as T
    x = torch.randn(5, 12, requires_grad=True)
    a, b, c = T.split_cols(x, [4, 2, 6])
    assert a.shape == (5, 4) and b.shape == (5, 2) and c.shape == (5, 6)
    assert a.data_ptr() == x.data_ptr() and c.stride(0) == 12      # views, no copies
    (2.0 * a).sum().backward(retain_graph=True)
    ref = torch.zeros(5, 12)
    ref[:, :4] = 2.0
    assert torch.equal(x.grad, ref)
    x.grad = None
    ((a * 1.5).sum() + (c * c).sum()).backward()
    ref = torch.zeros(5, 12)
    ref[:, :4] = 1.5
    ref[:, 6:] = 2.0 * x.detach()[:, 6:]
    assert torch.allclose(x.grad, ref)
    # without autograd: plain views
    y = torch.randn(3, 6)
    p, q = T.split_cols(y, [2, 4])
    assert p.data_ptr() == y.data_ptr() and q.shape == (3, 4)
