"""GPU parity of the fused SpatialTransformer entry (lr_stin_block_f16: proj_in + LayerNorm + fused q|k|v projection, C = 320).

Reference semantics: `x = self.proj_in(x)` (ldm/modules/attention.py:405-408, use_linear) followed by `self.norm1(x)` and attn1's
to_q / to_k / to_v on it (attention.py:280, 168-172); oracle: fp32 linear + unet_ref.layer_norm + fp32 linear on the fp16-rounded
inputs, with x1 rounded to fp16 between the two stages like the unfused path stores it.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402
from tests.test_gpu_ops import dev, h16, report  # noqa: E402

C = 320


def _params(tag, NQ):
    wp = h16(torch.from_numpy(weights.fill_like(f"stin.{tag}.proj_in.weight", (C, C))))
    bp = torch.from_numpy(weights.fill_like(f"stin.{tag}.proj_in.bias", (C,)))
    wq = h16(torch.from_numpy(weights.fill_like(f"stin.{tag}.qkv.weight", (NQ, C))))
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like(f"stin.{tag}.norm.weight", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like(f"stin.{tag}.norm.bias", (C,), kind="unit"))
    return wp, bp, wq, gamma, beta


def _run(x, wp, bp, wq, gamma, beta, dtype=torch.float16):
    from leftrefill_amd import ops, packing
    d = dev()
    wf, bf, _cs = packing.fold_layernorm(wq, None, gamma, beta, dtype)
    return ops.stin_block(x.to(dtype).to(d), wp.to(dtype).to(d), bp.to(d), wf.to(d), bf.to(d), eps=1e-5)


@pytest.mark.parametrize("M,NQ", [(256, 960), (512, 960), (1024, 192), (768, 384)])
def test_stin_block_vs_oracle(M, NQ):
    wp, bp, wq, gamma, beta = _params("p", NQ)
    x = h16(G.T(f"stin.{M}.{NQ}.x", (M, C)) * 1.2 + 0.1)
    x1_ref = F.linear(x, wp, bp)
    x1, qkv = _run(x, wp, bp, wq, gamma, beta)
    report(f"stin x1 M{M}", x1, x1_ref)
    # second stage against the oracle evaluated on the x1 the kernel stored (the reference path normalises the stored tensor)
    qkv_ref = F.linear(unet_ref.layer_norm(x1.float().cpu(), gamma, beta), wq)
    report(f"stin qkv M{M} NQ{NQ}", qkv, qkv_ref, atol=2e-3)
    # and end to end in fp32 (x1 never rounded): one more 16-bit hand-off than a single GEMM
    report(f"stin qkv(fp32 chain) M{M} NQ{NQ}", qkv, F.linear(unet_ref.layer_norm(x1_ref, gamma, beta), wq), atol=4e-3)


def test_stin_block_hot_shape_reruns_and_engine_path():
    """configs[1] level-0 shape (M = 8 x 8192): bit-identical reruns (the counted vmcnt waits cover stores and LDS-DMA loads alike),
    sampled rows against the oracle, and the two-GEMM engine path it replaces."""
    from leftrefill_amd import ops, packing
    d = dev()
    M, NQ = 65536, 960
    wp, bp, wq, gamma, beta = _params("hot", NQ)
    g = torch.Generator().manual_seed(11)
    x = h16(torch.randn(M, C, generator=g))
    outs = [_run(x, wp, bp, wq, gamma, beta) for _ in range(4)]
    for x1_, q_ in outs[1:]:
        assert torch.equal(x1_, outs[0][0]) and torch.equal(q_, outs[0][1])
    x1, qkv = outs[0]
    rows = torch.arange(0, M, 53)
    x1_ref = F.linear(x[rows], wp, bp)
    report("stin hot x1", x1[rows.to(d)], x1_ref)
    report("stin hot qkv", qkv[rows.to(d)], F.linear(unet_ref.layer_norm(x1[rows.to(d)].float().cpu(), gamma, beta), wq), atol=2e-3)
    # the launches it replaces: proj_in GEMM with row statistics, LayerNorm-folded q|k|v GEMM
    wf, bf, cs = packing.fold_layernorm(wq, None, gamma, beta)
    xd = x.half().to(d)
    y1, st = ops.gemm_conv(xd, wp.half().to(d), B=1, H=1, W=M, taps=1, bias=bp.to(d), want_stats=True)
    y2 = ops.gemm_conv(y1, wf.to(d), B=1, H=1, W=M, taps=1, bias=bf.to(d), ln=(st, 1e-5, cs.to(d)))
    d1 = (y1.float() - x1.float()).abs().max().item()
    print(f"[stin vs two GEMMs] x1 max abs diff {d1:.3e} (bit-equal: {torch.equal(y1, x1)})")
    assert d1 <= 2e-3 * max(1.0, y1.float().abs().max().item())
    err = (y2.float() - qkv.float()).abs().max().item()
    print(f"[stin vs two GEMMs] qkv max abs diff {err:.3e} at |qkv| {y2.float().abs().max().item():.2f}")
    assert err <= 4e-3 * max(1.0, y2.float().abs().max().item())


def test_stin_unsupported_shapes_are_reported():
    from leftrefill_amd import ops
    assert ops.stin_ok(65536, 320, 960)
    assert not ops.stin_ok(65536 + 128, 320, 960)      # ragged rows
    assert not ops.stin_ok(16384, 640, 1920)           # other widths keep the two GEMMs
    assert not ops.stin_ok(4096, 320, 1000)


def test_spatial_transformer_stin_equals_two_gemm_path():
    """engine.spatial_transformer at C = 320 with the fused entry vs the proj_in GEMM + LayerNorm-folded q|k|v GEMM."""
    import importlib
    from leftrefill_amd import engine
    from leftrefill_amd.dropin import install
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(3)
    d = dev()
    st = att.SpatialTransformer(320, 5, 64, depth=1, context_dim=1024, use_linear=True).to(d).eval()
    with torch.no_grad():
        for p_ in st.parameters():
            p_.copy_(torch.randn_like(p_) * 0.05)
        for n_ in (st.norm, st.transformer_blocks[0].norm1, st.transformer_blocks[0].norm2, st.transformer_blocks[0].norm3):
            n_.weight.add_(1.0)
    ps = engine.PackedST(st)
    N, H, W, Lc = 2, 16, 32, 77
    x = torch.randn(N * H * W, 320, device=d).half()
    ctx = torch.randn(N * Lc, 1024, device=d).half()
    outs = []
    for flag in (True, False):
        engine.STIN = flag
        try:
            with torch.no_grad():
                outs.append(engine.spatial_transformer(engine.Act(x, N, H, W), ctx, Lc, ps).tok.float().cpu())
        finally:
            engine.STIN = True
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"[stin vs two-GEMM entry] max abs diff {err:.3e} at |out| {outs[1].abs().max().item():.2f}")
    assert err <= 1e-2 * max(1.0, outs[1].abs().max().item())


def test_stin_block_with_groupnorm_equals_groupnorm_apply_then_stin_bitwise():
    """gn=...: the SpatialTransformer's GroupNorm (attention.py:399-404) applied to the rows inside the launch, from the producer's per-group
    partials -- the same arithmetic as lr_groupnorm_apply_n, so x1 and qkv are BIT-identical to the two-launch form; also through
    engine.spatial_transformer (LEFTREFILL_STIN_GN on / off)."""
    import importlib
    from leftrefill_amd import engine, ops, packing
    from leftrefill_amd.dropin import install
    d = dev()
    N, H, W = 2, 16, 32
    wp, bp, wq, gamma, beta = _params("gn", 960)
    gng = (1.0 + 0.3 * G.T("stin.gn.g", (C,))).to(d)
    gnb = (0.2 * G.T("stin.gn.b", (C,))).to(d)
    x0 = h16(G.T("stin.gn.x", (N, C, H, W)) * 0.7 + 0.4)
    conv_w = h16(torch.from_numpy(weights.fill_like("stin.gn.conv", (C, C, 3, 3))))
    tok = x0.permute(0, 2, 3, 1).reshape(N * H * W, C).half().contiguous().to(d)
    y, gs = ops.gemm_conv(tok, packing.pack_conv(conv_w).to(d), B=N, H=H, W=W, taps=9, want_gn_stats=True)
    part, R, gp, chunks = gs
    assert gp is not None, "the producer's plan must emit per-group partials for this shape"
    wf, bf, _cs = packing.fold_layernorm(wq, None, gamma, beta)
    args = (wp.half().to(d), bp.to(d), wf.to(d), bf.to(d))
    h = ops.group_norm_groups(y, N, H * W, gng, gnb, 1e-6, False, gp, chunks)
    x1_ref, qkv_ref = ops.stin_block(h, *args, eps=1e-5)
    x1, qkv = ops.stin_block(y, *args, eps=1e-5, gn=(gp, chunks, H * W, gng, gnb, 1e-6))
    assert torch.equal(x1, x1_ref) and torch.equal(qkv, qkv_ref)
    # the engine path
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(7)
    st = att.SpatialTransformer(320, 5, 64, depth=1, context_dim=1024, use_linear=True).to(d).eval()
    with torch.no_grad():
        for p_ in st.parameters():
            p_.copy_(torch.randn_like(p_) * 0.05)
        for n_ in (st.norm, st.transformer_blocks[0].norm1, st.transformer_blocks[0].norm2, st.transformer_blocks[0].norm3):
            n_.weight.add_(1.0)
    ps = engine.PackedST(st)
    ctx = torch.randn(N * 77, 1024, device=d).half()
    outs = []
    for flag in (True, False):
        engine.STIN_GN = flag
        try:
            with torch.no_grad():
                outs.append(engine.spatial_transformer(engine.Act(y, N, H, W, gs=gs), ctx, 77, ps).tok)
        finally:
            engine.STIN_GN = True
    assert torch.equal(outs[0], outs[1])
