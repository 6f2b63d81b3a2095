"""GPU parity of the row-resident LayerNorm + Linear at C = 640 (lr_rowlin_f16) against the CPU oracle.

Reference semantics: `self.norm1(x)` + the fused to_q / to_k / to_v of attn1 (ldm/modules/attention.py:280, 168-172) and `self.norm3(x)` +
GEGLU.proj + `x * F.gelu(gate)` (attention.py:282, 51-58); oracle: unet_ref.layer_norm + fp32 linear (+ exact-erf GELU gate) on the
fp16-rounded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402
from tests.test_gpu_ops import dev, h16, report  # noqa: E402

C = 640


def _params(tag, N, C=C):
    w = h16(torch.from_numpy(weights.fill_like(f"rowlin.{tag}.weight", (N, C))))
    b = torch.from_numpy(weights.fill_like(f"rowlin.{tag}.bias", (N,)))
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like(f"rowlin.{tag}.norm.weight", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like(f"rowlin.{tag}.norm.bias", (C,), kind="unit"))
    return w, b, gamma, beta


@pytest.mark.parametrize("M,N,C", [(128, 1920, 640), (256, 1920, 640), (384, 64, 640), (2048, 1920, 640), (1152, 320, 640),
                                   (128, 3840, 1280), (512, 1280, 1280), (256, 192, 1280), (4096, 1280, 1280)])
def test_rowlin_plain_vs_oracle(M, N, C):
    from leftrefill_amd import _lib, ops, packing
    if C == 1280 and not _lib.dev_variants():
        pytest.skip("the C = 1280 instance (measured, lost) is compiled in developer builds only")
    d = dev()
    w, b, gamma, beta = _params("p", N, C)
    x = h16(G.T(f"rowlin.{M}.{N}.{C}.x", (M, C)) * 1.3 + 0.2)
    ref = F.linear(unet_ref.layer_norm(x, gamma, beta), w, b)
    wf, bf, _cs = packing.fold_layernorm(w, b, gamma, beta)
    out = ops.rowlin(x.half().to(d), wf.to(d), bf.to(d), eps=1e-5)
    report(f"rowlin C{C} M{M} N{N}", out, ref, atol=2e-3 if C == 640 else 3e-3)


@pytest.mark.parametrize("M,C", [(256, 640), (2048, 640), (512, 1280), (4096, 1280)])
def test_rowlin_without_layernorm_is_the_plain_linear(M, C):
    """ln = 0: SpatialTransformer.proj_in (attention.py:405-408) -- bit-compared with the tiled GEMM too (same k order, one rounding)."""
    from leftrefill_amd import _lib, ops
    if C == 1280 and not _lib.dev_variants():
        pytest.skip("the C = 1280 instance (measured, lost) is compiled in developer builds only")
    d = dev()
    w, b, _g, _b = _params("lin", C, C)
    x = h16(G.T(f"rowlin.lin.{M}.{C}.x", (M, C)))
    out = ops.rowlin(x.half().to(d), w.half().to(d), b.to(d), ln=False)
    report(f"rowlin plain C{C} M{M}", out, F.linear(x, w, b), atol=2e-3)
    tiled = ops.gemm_conv(x.half().to(d), w.half().to(d), B=1, H=1, W=M, taps=1, bias=b.to(d))
    print(f"[rowlin ln=0 vs tiled GEMM C{C}] bit-equal: {torch.equal(out, tiled)}; max abs diff {(out.float() - tiled.float()).abs().max().item():.2e}")
    assert (out.float() - tiled.float()).abs().max().item() <= 2e-3


@pytest.mark.parametrize("M,H,C", [(128, 2560, 640), (512, 2560, 640), (256, 64, 640), (2048, 2560, 640), (256, 5120, 1280), (128, 96, 1280),
                                   (256, 1280, 320), (1024, 1280, 320), (512, 32, 320), (768, 96, 320)])
def test_rowlin_geglu_vs_oracle(M, H, C):
    """value * gelu(gate) of the interleaved projection (packing.geglu_perm), H output columns."""
    from leftrefill_amd import _lib, ops, packing
    if C != 640 and not _lib.dev_variants():
        pytest.skip("the C = 320 / 1280 instances (measured, lost) are compiled in developer builds only")
    d = dev()
    w, b, gamma, beta = _params("g", 2 * H, C)
    x = h16(G.T(f"rowlin.g.{M}.{H}.{C}.x", (M, C)) * 1.1 - 0.1)
    y = F.linear(unet_ref.layer_norm(x, gamma, beta), w, b)
    ref = y[:, :H] * F.gelu(y[:, H:])
    wf, bf, _cs = packing.fold_layernorm(w, b, gamma, beta)
    perm = packing.geglu_perm(H)
    out = ops.rowlin(x.half().to(d), wf[perm].contiguous().to(d), bf[perm].contiguous().to(d), eps=1e-5, geglu=True)
    # (the normalised rows are rounded to 16 bits before the product, and value x gate multiplies two such results: one element in 1.3 M
    # reached 5.2e-3 at |out| ~ 6)
    report(f"rowlin geglu C{C} M{M} H{H}", out, ref, atol=6e-3)


def test_rowlin_hot_shapes_reruns_and_tiled_gemm():
    """configs[1] level-1 shapes (M = 8 x 2048): bit-identical reruns (stores and LDS-DMA loads share the counted vmcnt waits), sampled
    rows against the oracle, and agreement with the LayerNorm-folded tiled GEMMs they replace."""
    from leftrefill_amd import ops, packing
    d = dev()
    M = 16384
    g = torch.Generator().manual_seed(21)
    x = h16(torch.randn(M, C, generator=g) * 1.2)
    xd = x.half().to(d)
    xf = x.float()
    st = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(M, 1, 2).contiguous().to(d)
    rows = torch.arange(0, M, 41)
    for geglu, N in ((False, 1920), (True, 5120)):
        w, b, gamma, beta = _params("hot%d" % geglu, N)
        wf, bf, cs = packing.fold_layernorm(w, b, gamma, beta)
        if geglu:
            perm = packing.geglu_perm(N // 2)
            wf, bf, cs = wf[perm].contiguous(), bf[perm].contiguous(), cs[perm].contiguous()
        wf, bf, cs = wf.to(d), bf.to(d), cs.to(d)
        outs = [ops.rowlin(xd, wf, bf, eps=1e-5, geglu=geglu) for _ in range(4)]
        assert all(torch.equal(o, outs[0]) for o in outs[1:])
        y = F.linear(unet_ref.layer_norm(x[rows], gamma, beta), w, b)
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:]) if geglu else y
        report(f"rowlin hot geglu={geglu}", outs[0][rows.to(d)], ref, atol=3e-3)
        tiled = ops.gemm_conv(xd, wf, B=1, H=1, W=M, taps=1, bias=bf, geglu=geglu, ln=(st, 1e-5, cs))
        err = (tiled.float() - outs[0].float()).abs().max().item()
        print(f"[rowlin vs tiled GEMM geglu={geglu}] max abs diff {err:.3e} at |out| {tiled.float().abs().max().item():.2f}")
        assert err <= 6e-3 * max(1.0, tiled.float().abs().max().item())


def test_rowlin_unsupported_shapes_are_reported():
    from leftrefill_amd import ops
    assert ops.rowlin_ok(16384, 640, 1920) and ops.rowlin_ok(2048, 640, 5120, "geglu") and ops.rowlin_ok(16384, 640, 640, "in")
    assert not ops.rowlin_ok(4096, 1280, 3840) and not ops.rowlin_ok(4096, 1280, 1280, "in") and not ops.rowlin_ok(16384, 640, 640, "q")
    assert not ops.rowlin_ok(16384 + 64, 640, 1920)      # ragged rows
    assert not ops.rowlin_ok(65536, 320, 960)            # other widths keep their own paths
    assert not ops.rowlin_ok(1024, 640, 1920)            # too few rows for the column slices to fill the chip
    assert not ops.rowlin_ok(4096, 1280, 10240, "geglu")


def test_transformer_block_rowlin_equals_tiled_path():
    """engine.transformer_block at C = 640 with the row-resident projections vs the LayerNorm-folded tiled GEMMs."""
    import importlib
    from leftrefill_amd import engine, ops
    from leftrefill_amd.dropin import install
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(4)
    d = dev()
    blk = att.BasicTransformerBlock(640, 10, 64, context_dim=1024).to(d).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn_like(p_) * 0.04)
        for n_ in (blk.norm1, blk.norm2, blk.norm3):
            n_.weight.add_(1.0)
    pt = engine.PackedTBlock(blk)
    B, L, Lc = 2, 256, 77
    x = torch.randn(B * L, 640, device=d).half()
    xf = x.float()
    st = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(B * L, 1, 2).contiguous()
    ctx = torch.randn(B * Lc, 1024, device=d).half()
    kv = ops.gemm_conv(ctx, pt.attn2.kv.w, B=1, H=1, W=B * Lc, taps=1)
    ent = (kv, None, ops.gemm_conv(ctx, pt.attn2.xk, B=1, H=1, W=B * Lc, taps=1), ops.xattn_pack_vt(kv[:, 640:], B, 10, Lc))
    outs = []
    for flag in (True, False):
        ops.ROWLIN = flag
        try:
            with torch.no_grad():
                outs.append(engine.transformer_block(x, ctx, pt, B, L, Lc, kv=ent, st=st)[0].float().cpu())
        finally:
            ops.ROWLIN = True
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"[row-resident vs tiled projections] max abs diff {err:.3e} at |out| {outs[1].abs().max().item():.2f}")
    assert err <= 1e-2 * max(1.0, outs[1].abs().max().item())


def test_rowlin_level0_geglu_hot_shape():
    """configs[1] level-0 feed-forward projection (M = 8 x 8192, C = 320, 2560 interleaved rows): reruns bit-identical, sampled rows vs oracle."""
    from leftrefill_amd import _lib, ops, packing
    if not _lib.dev_variants():
        pytest.skip("the C = 320 instance (measured, lost) is compiled in developer builds only")
    d = dev()
    M, C, H = 65536, 320, 1280
    g = torch.Generator().manual_seed(31)
    x = h16(torch.randn(M, C, generator=g) * 1.2)
    w, b, gamma, beta = _params("hot320", 2 * H, C)
    wf, bf, _cs = packing.fold_layernorm(w, b, gamma, beta)
    perm = packing.geglu_perm(H)
    wf, bf = wf[perm].contiguous().to(d), bf[perm].contiguous().to(d)
    xd = x.half().to(d)
    outs = [ops.rowlin(xd, wf, bf, eps=1e-5, geglu=True) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    rows = torch.arange(0, M, 67)
    y = F.linear(unet_ref.layer_norm(x[rows], gamma, beta), w, b)
    report("rowlin level-0 geglu", outs[0][rows.to(d)], y[:, :H] * F.gelu(y[:, H:]), atol=6e-3)


def test_spatial_transformer_split_feed_forward_equals_fused_block():
    """engine.spatial_transformer at C = 320: row-resident GEGLU projection + one composed GEMM (ops.FFN_SPLIT) vs the fused feed-forward block."""
    import importlib
    from leftrefill_amd import _lib, engine, ops
    if not _lib.dev_variants():
        pytest.skip("the split feed-forward (measured, lost) needs the C = 320 instance of developer builds")
    from leftrefill_amd.dropin import install
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(5)
    d = dev()
    st = att.SpatialTransformer(320, 5, 64, depth=1, context_dim=1024, use_linear=True).to(d).eval()
    with torch.no_grad():
        for p_ in st.parameters():
            p_.copy_(torch.randn_like(p_) * 0.05)
        for n_ in (st.norm, st.transformer_blocks[0].norm1, st.transformer_blocks[0].norm2, st.transformer_blocks[0].norm3):
            n_.weight.add_(1.0)
    ps = engine.PackedST(st)
    N, H, W, Lc = 2, 64, 128, 77          # 16384 rows: the split path wants whole 256-row blocks and enough of them
    x = torch.randn(N * H * W, 320, device=d).half()
    ctx = torch.randn(N * Lc, 1024, device=d).half()
    conv = torch.nn.Conv2d(320, 320, 3, padding=1).to(d)
    pc = engine.PackedConv(conv)
    outs = []
    prev_split = ops.FFN_SPLIT
    for flag in (True, False):
        ops.FFN_SPLIT = flag
        try:
            with torch.no_grad():
                act = engine.conv(engine.Act(x, N, H, W), pc, gn_stats=True)      # a producer with GroupNorm statistics, like a ResBlock
                outs.append(engine.spatial_transformer(act, ctx, Lc, ps).tok.float().cpu())
        finally:
            ops.FFN_SPLIT = prev_split
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"[split vs fused feed-forward] max abs diff {err:.3e} at |out| {outs[1].abs().max().item():.2f}")
    assert err <= 1e-2 * max(1.0, outs[1].abs().max().item())


def test_rowlin_with_groupnorm_equals_groupnorm_apply_then_linear_bitwise():
    """gn=...: SpatialTransformer.norm applied to the rows inside the launch (level 1: GroupNorm + proj_in), bit-identical to
    lr_groupnorm_apply_n followed by the plain row-resident Linear."""
    from leftrefill_amd import ops, packing
    d = dev()
    N, H, W, C = 2, 32, 64, 640
    w, b, _g, _b = _params("gnlin", C, C)
    gng = (1.0 + 0.3 * G.T("rowlin.gn.g", (C,))).to(d)
    gnb = (0.2 * G.T("rowlin.gn.b", (C,))).to(d)
    x0 = h16(G.T("rowlin.gn.x", (N, C, H, W)) * 0.7 + 0.4)
    conv_w = h16(torch.from_numpy(weights.fill_like("rowlin.gn.conv", (C, C, 3, 3))))
    tok = x0.permute(0, 2, 3, 1).reshape(N * H * W, C).half().contiguous().to(d)
    y, gs = ops.gemm_conv(tok, packing.pack_conv(conv_w).to(d), B=N, H=H, W=W, taps=9, want_gn_stats=True)
    part, R, gp, chunks = gs
    assert gp is not None
    h = ops.group_norm_groups(y, N, H * W, gng, gnb, 1e-6, False, gp, chunks)
    ref = ops.rowlin(h, w.half().to(d), b.to(d), ln=False)
    out = ops.rowlin(y, w.half().to(d), b.to(d), ln=False, gn=(gp, chunks, H * W, gng, gnb, 1e-6))
    assert torch.equal(out, ref)


def test_rowlin_4wave_blocks_developer_variant():
    """The measured-and-lost form (two independent 4-wave blocks per CU, knob LR_ROWLIN_W4, profiles/r06_rowlin_w4.txt): same bounds as the
    product kernel; the plain Linear is bit-equal to it (same k order per output, one rounding)."""
    from leftrefill_amd import _lib, ops, packing
    from tests.test_gpu_ops import need_dev_build
    need_dev_build()
    d = dev()
    M = 2048
    x = h16(G.T("rowlin.w4.x", (M, C)) * 1.3 + 0.2)
    w, b, gamma, beta = _params("p", 1920)
    wf, bf, _cs = packing.fold_layernorm(w, b, gamma, beta)
    wl, bl, _g, _b = _params("lin", C)
    wg, bg, gg, betag = _params("g", 5120)
    perm = packing.geglu_perm(2560)
    wgf, bgf, _ = packing.fold_layernorm(wg, bg, gg, betag)
    outs = {}
    try:
        for mode in (0, 1):
            _lib.dev_set("LR_ROWLIN_W4", mode)
            outs[mode] = (ops.rowlin(x.half().to(d), wf.to(d), bf.to(d), eps=1e-5),
                          ops.rowlin(x.half().to(d), wl.half().to(d), bl.to(d), ln=False),
                          ops.rowlin(x.half().to(d), wgf[perm].contiguous().to(d), bgf[perm].contiguous().to(d), eps=1e-5, geglu=True))
    finally:
        _lib.dev_set("LR_ROWLIN_W4", None)
    report("rowlin w4 q|k|v", outs[1][0], F.linear(unet_ref.layer_norm(x, gamma, beta), w, b), atol=2e-3)
    assert torch.equal(outs[1][1], outs[0][1])
    for a, bb in zip(outs[1], outs[0]):
        assert (a.float() - bb.float()).abs().max().item() <= 6e-3
