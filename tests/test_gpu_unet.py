"""GPU parity of the drop-in modules and the whole UNet against the reference goldens and the CPU oracle.

Whole-network tolerance.  The reference runs this path under fp16 autocast; rounding every conv/linear output to
fp16 moves the output of the 60-layer UNet by ~2e-3 relative (measured with the oracle's `autocast16` emulation
against its fp32 mode -- see test_oracle_golden.py).  No fp16 pipeline can therefore match the fp32 golden
element-wise at atol 1e-3; the whole-UNet criterion is:
  (1) relative L2 error vs the fp32 golden <= the autocast16-emulation's own error (we are at least as accurate as
      the reference's production numerics; no slack factor), and <= 4e-3 absolute cap;
  (2) the fraction of elements outside atol 1e-3 + rtol 2e-3 |ref| is no larger than the emulation's, max-abs within
      2x the emulation's, and HIP against the emulation is bounded by the sum of the two errors (assert_unet_row);
      every case's numbers are written to profiles/rNN_parity_table.txt by tools/parity_table.py.
Per-operator tests (test_gpu_ops.py) use the north-star rtol 2e-3 / atol 1e-3 directly.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402


def dev():
    return torch.device("cuda:0")


def install():
    import leftrefill_amd.dropin as dropin
    dropin.install()


def stats(name, out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"[{name}] shape {tuple(ref.shape)} max_abs {err.max().item():.3e} rel_l2 {rel:.3e} "
          f"ref_absmax {ref.abs().max().item():.3f}")
    assert torch.isfinite(out).all(), name
    return rel, err.max().item()


# ---------------------------------------------------------------------------------------------------------------
# composite operator goldens (G3) through the drop-in nn.Modules (reference constructor + state-dict keys)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kind,p", [c for c in G.OP_CASES if c[1] in ("res", "attn", "ff", "tblock", "st")],
                         ids=[c[0] for c in G.OP_CASES if c[1] in ("res", "attn", "ff", "tblock", "st")])
def test_module_goldens(golden, name, kind, p):
    install()
    from ldm.modules import attention as A
    from ldm.modules.diffusionmodules import openaimodel as O
    st = G.op_state(name, kind, p)
    i = {k: v.to(dev()) for k, v in G.op_inputs(name, kind, p).items()}
    if kind == "res":
        mod = O.ResBlock(p["Cin"], 1280, 0, out_channels=p["Cout"], dims=2, use_checkpoint=True)
    elif kind == "attn":
        mod = A.CrossAttention(p["C"], context_dim=p["ctx"], heads=p["heads"], dim_head=64)
    elif kind == "ff":
        mod = A.FeedForward(p["C"], glu=True)
    elif kind == "tblock":
        mod = A.BasicTransformerBlock(p["C"], p["heads"], 64, context_dim=p["ctx"])
    else:
        mod = A.SpatialTransformer(p["C"], p["heads"], 64, depth=1, context_dim=p["ctx"], use_linear=True)
    missing, unexpected = mod.load_state_dict(st, strict=True)
    assert not missing and not unexpected
    mod = mod.to(dev()).eval()
    with torch.no_grad():
        if kind == "res":
            y = mod(i["x"], i["emb"])
        elif kind == "attn":
            y = mod(i["x"], context=i.get("ctx"))
        elif kind == "ff":
            y = mod(i["x"])
        else:
            y = mod(i["x"], context=i["ctx"])
    ref = torch.from_numpy(golden("ops")[name])
    emul = G.op_oracle(name, kind, p, mode="autocast16")
    rel, mx = stats("module " + name, y, ref)
    rel_e, mx_e = stats("  autocast16-emulation " + name, emul, ref)
    assert rel <= max(1.5 * rel_e, 1e-3), (rel, rel_e)
    assert mx <= max(2.0 * mx_e, 4e-3), (mx, mx_e)


# ---------------------------------------------------------------------------------------------------------------
# whole UNet
# ---------------------------------------------------------------------------------------------------------------
_models = {}


def get_model(cname, multiview=None):
    install()
    key = (cname, multiview)
    if key in _models:
        return _models[key]
    if multiview is None:
        from ldm.modules.diffusionmodules.openaimodel import UNetModel
        cfg = G.CONFIGS[cname]
        m = UNetModel(**cfg.kwargs())
        sd = G.unet_state(cname)
    else:
        from ldm.modules.diffusionmodules.multiview_unet import MultiViewUnetModel
        cfg = G.mv_config(*multiview)
        m = MultiViewUnetModel(**cfg.kwargs())
        sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).eval()
    _models[key] = (m, sd, cfg)
    return _models[key]


def measure_unet(case, cname, N, H, W, ts, ref=None, multiview=None, bisect=False, emulate=True):
    """Runs one whole-UNet case eagerly and through the hipGraph and returns its parity metrics against `ref` (the reference
    golden; None: the oracle's fp32 forward, which is pinned to the reference by tests/test_oracle_golden.py):
    rel-L2 / max-abs / fraction of elements outside the north-star tolerance (rtol 2e-3, atol 1e-3), for the HIP path and
    for the oracle's emulation of the reference's own fp16-autocast numerics, plus HIP against that emulation."""
    m, sd, cfg = get_model(cname, multiview)
    x, t, ctx = G.unet_inputs(case, cfg, N, H, W, ts)
    taps = {}
    m.use_hip_graph = False
    if bisect:
        m.__dict__["_lr_taps"] = taps
    with torch.no_grad():
        y_eager = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
    m.__dict__.pop("_lr_taps", None)
    m.use_hip_graph = True
    with torch.no_grad():
        y_graph = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
        y_graph2 = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
    assert y_eager.dtype == torch.float16
    assert torch.equal(y_eager, y_graph), "hipGraph replay must be bit-identical to eager launches"
    assert torch.equal(y_graph, y_graph2), "replays must be bit-identical"
    otaps = {} if bisect else None
    if ref is None:
        ref = unet_ref.unet_forward(sd, cfg, x, t, ctx, taps=otaps)
    elif bisect:
        unet_ref.unet_forward(sd, cfg, x, t, ctx, taps=otaps)
    assert y_eager.shape == ref.shape
    rel, mx = stats("unet " + case, y_eager, ref)
    row = dict(case=case, shape=tuple(ref.shape), rel=rel, max_abs=mx, viol=viol_frac(y_eager, ref), ref_absmax=ref.abs().max().item())
    if emulate:
        emul = unet_ref.unet_forward(sd, cfg, x, t, ctx, mode="autocast16")
        rel_e, mx_e = stats("  autocast16-emulation " + case, emul, ref)
        row.update(rel_emu=rel_e, max_abs_emu=mx_e, viol_emu=viol_frac(emul, ref), viol_hip_vs_emu=viol_frac(y_eager, emul),
                   rel_hip_vs_emu=((y_eager.float().cpu() - emul).norm() / emul.norm()).item())
        print(f"    north-star rtol 2e-3 / atol 1e-3 violations: HIP vs fp32 {100 * row['viol']:.3f} %  autocast16-emulation vs "
              f"fp32 {100 * row['viol_emu']:.3f} %  HIP vs emulation {100 * row['viol_hip_vs_emu']:.3f} %  ({ref.numel()} elements)")
    if bisect:
        for k in otaps:
            e = (taps[k].cpu() - otaps[k]).norm() / otaps[k].norm()
            print(f"    tap {k:6s} rel_l2 {e.item():.3e}")
    return row


def assert_unet_row(row):
    """Whole-network criterion, no slack factors (restored in round 5 after VERDICT r4 #2 / ADVICE r4).  Against the fp32
    reference the HIP path is at least as accurate as the reference's own fp16-autocast numerics (oracle emulation) in relative L2
    and in the fraction of elements outside the north-star tolerance; its largest error stays within 2x the emulation's (a max over
    ~1e5..1e6 elements is a noisy statistic); and HIP against the emulation is bounded too (two fp16 pipelines around the same fp32
    answer differ by at most the sum of their errors).  A change that needs a band on the first line again is a parity regression."""
    rel, rel_e = row["rel"], row["rel_emu"]
    assert rel <= min(rel_e, 4e-3), (rel, rel_e)
    assert row["max_abs"] <= max(2.0 * row["max_abs_emu"], 5e-3), (row["max_abs"], row["max_abs_emu"])
    assert row["viol"] <= row["viol_emu"], (row["viol"], row["viol_emu"])
    assert row["rel_hip_vs_emu"] <= rel + rel_e, (row["rel_hip_vs_emu"], rel, rel_e)


def check_unet(golden, case, cname, N, H, W, ts, fname="unet", multiview=None, bisect=True):
    ref = torch.from_numpy(golden(fname)[case])
    row = measure_unet(case, cname, N, H, W, ts, ref=ref, multiview=multiview, bisect=bisect)
    assert_unet_row(row)
    return row


def viol_frac(out, ref, rtol=2e-3, atol=1e-3):
    out, ref = out.float().cpu(), ref.float().cpu()
    return ((out - ref).abs() > atol + rtol * ref.abs()).float().mean().item()


_DIGEST_SNIPPET = """
import hashlib, sys, torch
sys.path.insert(0, {root!r})
import tests.test_gpu_unet as T
from oracle import golden_spec as G
case, cname, N, H, W, ts = [c for c in G.UNET_CASES if c[1] == "MID"][0]
m, sd, cfg = T.get_model(cname)
x, t, ctx = G.unet_inputs(case, cfg, N, H, W, ts)
with torch.no_grad():
    y = m(x.cuda(), t.cuda(), ctx.cuda())
print("DIGEST", hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
"""


def test_two_fresh_processes_are_bit_identical():
    """The tile / split-K plan is a pure function of the shape (in-tree table + static heuristic): two fresh processes
    produce bit-identical eps on the MID config, and so does this process (what the sharded multi-view path relies on
    for its replicated rows -- leftrefill_amd/dist.py)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", _DIGEST_SNIPPET.format(root=root)], capture_output=True, text=True,
                             cwd=root, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    case, cname, N, H, W, ts = [c for c in G.UNET_CASES if c[1] == "MID"][0]
    m, sd, cfg = get_model(cname)
    x, t, ctx = G.unet_inputs(case, cfg, N, H, W, ts)
    with torch.no_grad():
        y = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
    digests.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
    assert digests[0] == digests[1] == digests[2], digests


@pytest.mark.parametrize("case,cname,N,H,W,ts", [c for c in G.UNET_CASES if c[1] == "MID"],
                         ids=[c[0] for c in G.UNET_CASES if c[1] == "MID"])
def test_unet_mid(golden, case, cname, N, H, W, ts):
    check_unet(golden, case, cname, N, H, W, ts)


@pytest.mark.parametrize("case,cname,N,H,W,ts", [c for c in G.UNET_CASES if c[1] == "FULL"],
                         ids=[c[0] for c in G.UNET_CASES if c[1] == "FULL"])
def test_unet_full_width(golden, case, cname, N, H, W, ts):
    """The shipped 866 M-parameter SD2-inpainting UNet config against the reference golden."""
    check_unet(golden, case, cname, N, H, W, ts, bisect=False)


@pytest.mark.parametrize("case,V,concat,b,H,W", G.MV_CASES, ids=[c[0] for c in G.MV_CASES])
def test_multiview_unet(golden, case, V, concat, b, H, W):
    n = b * (V - 1 if concat else V)
    check_unet(golden, case, "MV", n, H, W, [501] * n, fname="multiview", multiview=(V, concat))


def test_multiview_canvas_sharded_path_single_rank(golden):
    """`mv_shard=True` code path (all-gather of canvases + own-row attention) in its 1-rank form == the fused path."""
    case, V, concat, b, H, W = [c for c in G.MV_CASES if c[0] == "mv_v2_concat"][0]
    m, sd, cfg = get_model("MV", (V, concat))
    n = b * (V - 1)
    x, t, ctx = G.unet_inputs(case, cfg, n, H, W, [501] * n)
    ref = torch.from_numpy(golden("multiview")[case])
    with torch.no_grad():
        y_fused = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
        m.mv_shard = True
        try:
            y_shard = m(x.to(dev()), t.to(dev()), ctx.to(dev()))
        finally:
            m.mv_shard = False
    rel, _ = stats("mv sharded (1 rank) " + case, y_shard, ref)
    assert torch.equal(y_shard, y_fused), "1-rank sharded path must reproduce the fused gather/scatter path bit for bit"
    assert rel < 4e-3


@pytest.mark.parametrize("split", [True, False], ids=["split_target", "replicated_target"])
def test_multiview_sharded_block_row_copy_glue_equals_torch_glue(monkeypatch, split):
    """The sharded block's glue (pack rows + LayerNorm statistics into one message, unpack into sequence order and own rows, write the
    canvas back) as three lr_row_copy launches == the torch slice / cat form that the gloo tests pin against the oracle, bit for bit:
    one rank of a simulated 4-rank job (LEFTREFILL_MV_SIM_WORLD: the peers' rows are copies of the local ones), every rank id, with
    the target rows split over the ranks and replicated."""
    from leftrefill_amd import engine
    V = 5      # 4 ranks: the target rows (256 / 64 / 16 / 4 per level at this size) split evenly
    m, sd, cfg = get_model("MV", (V, True))
    x, t, ctx = G.unet_inputs("mv_glue", cfg, 2, 16, 32, [501, 501])
    monkeypatch.setenv("LEFTREFILL_MV_SIM_WORLD", str(V - 1))
    monkeypatch.setattr(engine, "MV_SPLIT_TARGET", split)
    m.mv_shard = True
    try:
        for rank in range(V - 1):
            monkeypatch.setenv("LEFTREFILL_MV_SIM_RANK", str(rank))
            outs = []
            for rc in (True, False):
                monkeypatch.setattr(engine, "MV_ROW_COPY", rc)
                with torch.no_grad():
                    outs.append(m(x.to(dev()), t.to(dev()), ctx.to(dev())))
            assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), f"rank {rank}"
    finally:
        m.mv_shard = False


def test_context_kv_cache_is_invalidated_correctly():
    """The step graph caches the cross-attention K/V projections per context tensor (constant over the DDIM loop)."""
    m, sd, cfg = get_model("MID")
    x, t, ctx = G.unet_inputs("kvcache", cfg, 2, 16, 32, [981, 1])
    x, t = x.to(dev()), t.to(dev())
    ctx_a = ctx.to(dev())
    ctx_b = (ctx * 0.5 + 0.1).to(dev())
    m.use_hip_graph = False
    with torch.no_grad():
        ref_a, ref_b = m(x, t, ctx_a), m(x, t, ctx_b)
    m.use_hip_graph = True
    with torch.no_grad():
        a1 = m(x, t, ctx_a)
        a2 = m(x, t, ctx_a)            # cache hit (same tensor object, same version)
        b1 = m(x, t, ctx_b)            # different tensor -> recomputed
        ctx_a.copy_(ctx_b)             # in-place modification bumps _version -> recomputed
        a3 = m(x, t, ctx_a)
    assert torch.equal(a1, ref_a) and torch.equal(a2, ref_a)
    assert torch.equal(b1, ref_b) and torch.equal(a3, ref_b)
    assert not torch.equal(ref_a, ref_b)


def test_unet_full_width_headline_resolution_vs_oracle():
    """the shipped 866 M-parameter UNet at the HEADLINE resolution (configs[1]: latent 64x128) with N = 2,
    HIP vs the CPU oracle's fp32 forward (~25 s on the GPU box's host cores) and vs its fp16-autocast emulation -- the same
    criterion as every other whole-network case.  (The oracle restatement is pinned to the real reference by the goldens of
    tests/test_oracle_golden.py, incl. this config at 8x16 / 16x32.)"""
    row = measure_unet("unet_full_64x128_headline", "FULL", 2, 64, 128, [981, 21])
    assert_unet_row(row)


def test_multiview_mv5_full_size_sample_vs_oracle():
    """One mv5 sample of BASELINE configs[3] at full size: view_num = 5, concat_target (4 canvases [ref_i | target] of latent
    64x128, re-arranged cross-view self-attention over 5 x 4096 = 20 480 tokens), full width, HIP vs the oracle's fp32 forward."""
    V, concat = 5, True
    install()
    from ldm.modules.diffusionmodules.multiview_unet import MultiViewUnetModel
    cfg = unet_ref.UNetConfig(multiview=True, view_num=V, concat_target=concat)      # shipped widths (320, 64-channel heads, ctx 1024)
    key = ("MV_FULL", (V, concat))
    if key not in _models:
        m = MultiViewUnetModel(**cfg.kwargs())
        sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MVF.")
        m.load_state_dict(sd, strict=True)
        _models[key] = (m.to(dev()).eval(), sd, cfg)
    row = measure_unet("mv5_full_64x128", "MV_FULL", 4, 64, 128, [501] * 4, multiview=(V, concat), emulate=False)
    assert row["rel"] <= 4e-3 and row["viol"] <= 0.2, row


def test_full_size_properties_config2():
    """BASELINE configs[1] shape (full width, latent 64x128, UNet batch 8): size-independent properties on top of the direct
    oracle comparison at this resolution (test_unet_full_width_headline_resolution_vs_oracle): reruns and hipGraph replay are
    bit-identical; the network is equivariant to a permutation of the batch (bit-exact: every row / sample is computed
    independently of its position); the two halves of the batch run separately agree with the joint run to fp16 noise
    (tile / split-K choices are functions of M, so not bit-exact)."""
    m, sd, cfg = get_model("FULL")
    N, H, W = 8, 64, 128
    x, t, ctx = G.unet_inputs("full_size", cfg, N, H, W, [981, 801, 601, 401, 201, 101, 21, 1])
    d = dev()
    x, t, ctx = x.to(d), t.to(d), ctx.to(d)
    with torch.no_grad():
        m.use_hip_graph = True
        y = m(x, t, context=ctx)
        assert y.shape == (N, 4, H, W) and torch.isfinite(y).all()
        assert torch.equal(m(x, t, context=ctx), y)                      # graph replay, rerun
        m.use_hip_graph = False
        assert torch.equal(m(x, t, context=ctx), y)                      # eager launches == captured graph
        perm = torch.tensor([3, 7, 0, 5, 1, 6, 2, 4], device=d)
        assert torch.equal(m(x[perm], t[perm], context=ctx[perm]), y[perm])
        halves = torch.cat([m(x[:4], t[:4], context=ctx[:4]), m(x[4:], t[4:], context=ctx[4:])])
        m.use_hip_graph = True
    rel = ((halves.float() - y.float()).norm() / y.float().norm()).item()
    print(f"[full size 64x128 N=8] |y| max {y.abs().max().item():.3f}; halves vs joint rel_l2 {rel:.3e}")
    assert rel < 4e-3


@pytest.mark.parametrize("cname,N,H,W", [("MID", 4, 16, 32), ("FULL", 4, 16, 32), ("FULL", 8, 64, 128)],
                         ids=["mid_16x32", "full_16x32", "full_64x128_config2"])
def test_cfg_shared_prefix_is_exact(cname, N, H, W):
    """Classifier-free guidance runs the UNet on [x; x] with contexts [uncond; cond] (ddim.py:317-342).  With
    `cfg_shared_prefix` the context-free prefix (conv_in, first ResBlock, first self-attention, first cross-attention's query
    projection) runs once for both halves: the result must equal the full computation bit for bit (the half-batch GEMMs take
    the plan of the full batch, so every partial sum is formed in the same order), captured graph and eager alike."""
    m, sd, cfg = get_model(cname)
    d = dev()
    x, t, ctx = G.unet_inputs(f"cfgp_{cname}_{H}", cfg, N, H, W, [701] * N)
    x = torch.cat([x[: N // 2]] * 2).to(d)
    t = torch.cat([t[: N // 2]] * 2).to(d)
    ctx = ctx.to(d)                                   # N different contexts
    with torch.no_grad():
        for graph in (True, False):
            m.use_hip_graph = graph
            m.cfg_shared_prefix = False
            full = m(x, t, context=ctx)
            m.cfg_shared_prefix = True
            shared = m(x, t, context=ctx)
            m.cfg_shared_prefix = False
            assert torch.isfinite(full).all()
            same = (shared == full).float().mean().item()
            rel = ((shared.float() - full.float()).norm() / full.float().norm()).item()
            print(f"[cfg shared prefix {cname} N={N} {H}x{W} graph={graph}] identical elements {100 * same:.3f} %, rel_l2 {rel:.2e}")
            assert torch.equal(shared, full)
        m.use_hip_graph = True
    # halves that differ must not be served by the shared path: the flag is the caller's promise, so only check the promise
    # is what the sampler establishes (tests/test_gpu_sampler.py::test_sampler_shared_prefix_matches_full_cfg)


def test_full_size_properties_config4_mv5():
    """BASELINE configs[3] at full size: the shipped width, view_num = 5 with concat_target (4 canvases [ref_i | target] per
    sample at latent 64x128, re-arranged self-attention sequence 5 x 4096 = 20480 tokens), 2 samples = UNet batch 8.  Too
    large for the CPU oracle, so size-independent properties: finite output; hipGraph replay == rerun == eager launches
    bit-exactly; bit-exact equivariance to swapping the two samples; swapping two REFERENCE canvases of a sample permutes
    that sample's outputs up to fp16 noise (softmax over the same keys in another order)."""
    install()
    from ldm.modules.diffusionmodules.multiview_unet import MultiViewUnetModel
    cfg = unet_ref.UNetConfig(multiview=True, view_num=5, concat_target=True)
    sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV5.")
    m = MultiViewUnetModel(**cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).eval()
    N, H, W = 8, 64, 128
    x, t, ctx = G.unet_inputs("mv5_full", cfg, N, H, W, [801] * 4 + [201] * 4)
    d = dev()
    x, t, ctx = x.to(d), t.to(d), ctx.to(d)
    with torch.no_grad():
        m.use_hip_graph = True
        y = m(x, t, context=ctx)
        assert y.shape == (N, 4, H, W) and torch.isfinite(y).all()
        assert torch.equal(m(x, t, context=ctx), y)
        m.use_hip_graph = False
        assert torch.equal(m(x, t, context=ctx), y)
        swap = torch.tensor([4, 5, 6, 7, 0, 1, 2, 3], device=d)
        assert torch.equal(m(x[swap], t[swap], context=ctx[swap]), y[swap])
        refswap = torch.tensor([0, 2, 1, 3, 4, 5, 6, 7], device=d)        # reference canvases 1 and 2 of sample 0 trade places
        # (canvas 0 stays: its right half is the target the re-arranged sequence starts with, multiview_attention.py:436-448)
        y2 = m(x[refswap], t[refswap], context=ctx[refswap])
        m.use_hip_graph = True
    rel = ((y2[refswap].float() - y.float()).norm() / y.float().norm()).item()
    print(f"[mv5 full size] |y| max {y.abs().max().item():.3f}; reference-canvas swap rel_l2 {rel:.3e}")
    assert rel < 4e-3
    del m
    torch.cuda.empty_cache()


def _mv_shard_worker(rank, world, port, q, backend="gloo"):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend.startswith("nccl"):          # one process per GPU, RCCL over xGMI
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        install()
        from ldm.modules.diffusionmodules.multiview_unet import MultiViewUnetModel
        V, b, H, W = world + 1, 1, 8, 16
        cfg = G.mv_config(V, True)
        sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
        m = MultiViewUnetModel(**cfg.kwargs())
        m.load_state_dict(sd, strict=True)
        d = torch.device("cuda", rank) if backend.startswith("nccl") else dev()
        m = m.to(d).eval()
        m.mv_shard = True
        m.mv_shard_graph = backend == "nccl-graph"
        x, t, ctx = G.unet_inputs("mv_shard2", cfg, b * world, H, W, [501] * (b * world))
        sl = slice(rank, rank + 1)                     # this rank's canvas [ref_rank | target]
        with torch.no_grad():
            y = m(x[sl].to(d), t[sl].to(d), ctx[sl].to(d))
            if m.mv_shard_graph:
                assert torch.equal(m(x[sl].to(d), t[sl].to(d), ctx[sl].to(d)), y)      # replay of the captured step
        q.put((rank, y.float().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for an RCCL (nccl backend) group")
def test_multiview_canvas_sharded_over_rccl():
    """The same one-canvas-per-rank run on two real GPUs over the `nccl` backend (RCCL / xGMI): all_gather_into_tensor of
    the reference halves + broadcast of rank 0's target half per block; skipped on single-GPU boxes."""
    _run_mv_sharded("nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for an RCCL (nccl backend) group")
def test_multiview_canvas_sharded_over_rccl_in_hipgraph():
    """... with the whole step, collectives included, captured into a hipGraph on every rank (`mv_shard_graph`)."""
    _run_mv_sharded("nccl-graph")


def test_multiview_canvas_sharded_two_ranks_on_one_gpu():
    """One canvas per rank (world = view_num - 1 = 2; both ranks share cuda:0 and exchange over gloo because a 1-GPU box has
    no second RCCL peer): every transformer block gathers the reference halves and rank 0's target half, builds K/V for
    [target, ref_0, ref_1] and attends only for its own rows.  Each rank's canvas must match the CPU oracle of the joint
    multi-view UNet."""
    _run_mv_sharded("gloo")


def _run_mv_sharded(backend):
    import socket
    import torch.multiprocessing as mp
    world, H, W = 2, 8, 16
    V = world + 1
    cfg = G.mv_config(V, True)
    sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
    x, t, ctx = G.unet_inputs("mv_shard2", cfg, world, H, W, [501] * world)
    ref = unet_ref.unet_forward(sd, cfg, x, t, ctx)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_mv_shard_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
    for r in range(world):
        out = torch.from_numpy(res[r])
        rel = ((out - ref[r:r + 1]).norm() / ref[r:r + 1].norm()).item()
        print(f"[mv sharded 2 ranks] rank {r}: rel_l2 {rel:.3e}")
        assert rel < 4e-3
