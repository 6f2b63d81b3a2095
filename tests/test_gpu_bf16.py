"""bfloat16 instances of every kernel on the path (BASELINE configs[4] names bf16): the `_bf16` twin entry points of
include/leftrefill_hip.h against the same fp32 references the fp16 tests use, on inputs rounded to bf16 first so both
sides evaluate the same function.

Tolerance: the kernels accumulate / normalise / softmax in fp32 exactly like the fp16 instances; the only difference is
the 8-bit significand of the stored results (unit roundoff 2^-9 = 1.95e-3, vs 2^-12 for fp16).  The bound used here is
the north star's fp16 bound (rtol 2e-3 / atol 1e-3) scaled by that ratio of 8: rtol 1.6e-2, atol 8e-3 -- per element, on
outputs of O(1) magnitude -- and 3x tighter in relative L2 (printed for every case)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402

BF = torch.bfloat16
RTOL, ATOL, REL_L2 = 1.6e-2, 8e-3, 6e-3


def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def b16(x):
    return x.to(BF).float()


def to_tok(x):
    N, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(N * H * W, C).to(BF).contiguous().to(dev())


def from_tok(y, N, H, W):
    return y.float().cpu().reshape(N, H, W, -1).permute(0, 3, 1, 2)


def report(name, out, ref, rtol=RTOL, atol=ATOL, rel_l2=REL_L2):
    assert out.dtype in (BF, torch.float32), (name, out.dtype)
    out, ref = out.float().cpu(), ref.float().cpu()
    err = (out - ref).abs()
    bad = (err > atol + rtol * ref.abs()).sum().item()
    rel = (err.norm() / ref.norm()).item()
    print(f"[bf16 {name}] max_abs_err {err.max().item():.3e} rel_l2 {rel:.3e} viol {bad}/{err.numel()}")
    assert torch.isfinite(out).all(), name
    assert bad == 0 and rel <= rel_l2, f"{name}: {bad} outside rtol={rtol} atol={atol}; rel_l2 {rel:.3e}"


def test_layout_and_embedding_bf16():
    from leftrefill_amd import ops
    d = dev()
    x = G.T("bf.lay", (2, 9, 6, 10))
    tok = ops.nchw_to_nhwc(x.to(d), cpad=64, dtype=BF)
    assert tok.dtype == BF and tok.shape == (120, 64)
    assert torch.equal(tok[:, :9].cpu(), x.permute(0, 2, 3, 1).reshape(120, 9).to(BF)) and not tok[:, 9:].any()
    assert torch.equal(ops.nhwc_to_nchw(tok, 2, 6, 10, 9).cpu(), x.to(BF))
    assert torch.equal(ops.nhwc_to_nchw(tok, 2, 6, 10, 9, torch.float32).cpu(), b16(x))
    t = torch.tensor([0, 1, 501, 999], device=d)
    e = ops.timestep_embedding(t, 320, BF)
    report("timestep_embedding", e, unet_ref.timestep_embedding(t.cpu(), 320))


@pytest.mark.parametrize("C1,C2,N,H,W,silu", [(320, 0, 2, 8, 16, True), (640, 320, 2, 8, 8, True), (128, 0, 1, 16, 16, False)])
def test_groupnorm_layernorm_bf16(C1, C2, N, H, W, silu):
    from leftrefill_amd import ops
    d = dev()
    C = C1 + C2
    x = b16(G.T(f"bf.gn.{C1}.{C2}", (N, C, H, W)) * 1.5 + 0.3)
    g = torch.from_numpy(weights.fill_like(f"bf.gn.{C}.weight", (C,)))
    b = torch.from_numpy(weights.fill_like(f"bf.gn.{C}.bias", (C,)))
    ref = F.group_norm(x, 32, g, b, 1e-5)
    ref = F.silu(ref) if silu else ref
    y = ops.group_norm(to_tok(x[:, :C1]), N, H * W, g.to(d), b.to(d), 1e-5, silu, to_tok(x[:, C1:]) if C2 else None)
    report(f"groupnorm {C1}+{C2}", from_tok(y, N, H, W), ref)
    xl = b16(G.T(f"bf.ln.{C}", (N * H * W, C)))
    yl = ops.layer_norm(xl.to(BF).to(d), g.to(d), b.to(d), 1e-5)
    report(f"layernorm {C}", yl, F.layer_norm(xl, (C,), g, b, 1e-5))


def _conv_case(name, N, Cin, Cout, H, W, taps=9, stride=1, up=0, C2=0, rowvec=False, resid=False, tile_m=0, tile_n=0, splits=0, pipe=0):
    from leftrefill_amd import ops, packing
    d = dev()
    Ct = Cin + C2
    Hs, Ws = (2 * H, 2 * W) if stride == 2 else (H // 2, W // 2) if up else (H, W)
    k = 3 if taps == 9 else 1
    x = b16(G.T(name + ".x", (N, Ct, Hs, Ws)))
    w = b16(torch.from_numpy(weights.fill_like(name + ".w", (Cout, Ct, k, k))))
    b = torch.from_numpy(weights.fill_like(name + ".b", (Cout,)))
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1 if taps == 9 else 0)
    rv = rs = None
    if rowvec:
        rv = b16(G.T(name + ".rv", (N, Cout)))
        ref = ref + rv[:, :, None, None]
    if resid:
        rs = b16(G.T(name + ".rs", (N, Cout, H, W)))
        ref = ref + rs
    wp = packing.pack_conv(w, cin_pad=Ct, dtype=BF).to(d)
    assert wp.dtype == BF
    y = ops.gemm_conv(to_tok(x[:, :Cin]), wp, B=N, H=H, W=W, Hs=Hs, Ws=Ws, taps=taps, stride=stride, up=up,
                      x2=to_tok(x[:, Cin:]) if C2 else None, bias=packing.pack_bias(b).to(d),
                      rowvec=rv.to(BF).to(d) if rowvec else None, resid=to_tok(rs) if resid else None,
                      tile_m=tile_m, tile_n=tile_n, splits=splits, pipe=pipe)
    assert y.dtype == BF
    report(name, from_tok(y[:, :Cout], N, H, W), ref)


def test_gemm_conv_bf16_variants():
    _conv_case("bf.lin", 1, 320, 640, 10, 30, taps=1)
    _conv_case("bf.lin_res", 2, 640, 640, 8, 8, taps=1, resid=True)
    _conv_case("bf.c3", 2, 320, 320, 12, 20, rowvec=True)
    _conv_case("bf.c3_cat", 1, 640, 320, 8, 16, C2=320, resid=True)
    _conv_case("bf.s2", 2, 320, 320, 6, 10, stride=2)
    _conv_case("bf.up", 1, 640, 640, 8, 12, up=1)
    _conv_case("bf.out", 2, 320, 64, 8, 16)
    _conv_case("bf.splitk", 1, 1280, 1280, 8, 8, splits=4)
    # the halo-tile conv (conv_halo.hip) in bf16: both instances, concat + row vector + residual, split-K, the two-sample tile of 8-line images
    for tn in (160, 320):
        _conv_case(f"bf.halo{tn}", 2, 320, 320, 32, 16, C2=320, rowvec=True, resid=True, tile_m=256, tile_n=tn, splits=1, pipe=8)
        _conv_case(f"bf.halo{tn}_splitk", 4, 1280, 640, 8, 16, resid=True, tile_m=256, tile_n=tn, splits=3, pipe=8)


@pytest.mark.parametrize("tm,tn", [(128, 64), (128, 128), (128, 160), (256, 128), (256, 160), (256, 256), (256, 320)])
def test_gemm_tiles_bf16_agree_bitwise(tm, tn):
    """Every tile instance computes the same fp32 sums in the same K order: bf16 outputs are bit-identical across tiles."""
    from leftrefill_amd import ops, packing
    d = dev()
    x = to_tok(b16(G.T("bf.tiles.x", (2, 320, 16, 24))))
    w = packing.pack_conv(b16(torch.from_numpy(weights.fill_like("bf.tiles.w", (640, 320, 3, 3)))), dtype=BF).to(d)
    ref = ops.gemm_conv(x, w, B=2, H=16, W=24, taps=9, tile_m=128, tile_n=64, splits=1)
    y = ops.gemm_conv(x, w, B=2, H=16, W=24, taps=9, tile_m=tm, tile_n=tn, splits=1)
    assert torch.equal(ref, y)


def test_layernorm_fold_and_geglu_bf16():
    """LayerNorm folded into the consumer GEMM (row statistics from the producer's epilogue) and the GEGLU epilogue."""
    from leftrefill_amd import ops, packing
    d = dev()
    M, C = 384, 320
    x0 = b16(G.T("bf.lnf.x0", (M, C)))
    w0 = b16(torch.from_numpy(weights.fill_like("bf.lnf.w0", (C, C))))
    y0, st = ops.gemm_conv(x0.to(BF).to(d), w0.to(BF).to(d), B=1, H=1, W=M, taps=1, want_stats=True)
    h = y0.float().cpu()                                   # the bf16 activations the consumer reads
    report("producer", y0, F.linear(x0, w0))
    g = torch.from_numpy(weights.fill_like("bf.lnf.g", (C,))) + 1.0
    be = torch.from_numpy(weights.fill_like("bf.lnf.be", (C,)))
    w1 = torch.from_numpy(weights.fill_like("bf.lnf.w1", (3 * C, C)))
    b1 = torch.from_numpy(weights.fill_like("bf.lnf.b1", (3 * C,)))
    wf, bf, cs = packing.fold_layernorm(w1, b1, g, be, dtype=BF)
    assert wf.dtype == BF
    y1 = ops.gemm_conv(y0, wf.to(d), B=1, H=1, W=M, taps=1, bias=bf.to(d), ln=(st, 1e-5, cs.to(d)))
    ref = F.linear(F.layer_norm(h, (C,), g, be, 1e-5), w1, b1)
    # the folded form rounds W*gamma (not LN(x)) to bf16: a different, equally valid rounding of the same product
    report("ln-folded linear", y1, ref, rel_l2=8e-3, atol=2.5e-2)
    # GEGLU projection with and without the fold
    wg = torch.from_numpy(weights.fill_like("bf.gg.w", (8 * C, C)))
    bg = torch.from_numpy(weights.fill_like("bf.gg.b", (8 * C,)))
    wp, bp = packing.pack_geglu(b16(wg), bg, dtype=BF)
    xin = b16(F.layer_norm(h, (C,), g, be, 1e-5))
    u, gate = F.linear(xin, b16(wg), bg).chunk(2, dim=-1)
    yg = ops.gemm_conv(xin.to(BF).to(d), wp.to(d), B=1, H=1, W=M, taps=1, bias=bp.to(d), geglu=True)
    report("geglu", yg, u * F.gelu(gate))
    wff, bff, csf = packing.fold_layernorm(wg, bg, g, be, dtype=BF)
    perm = packing.geglu_perm(4 * C)
    yf = ops.gemm_conv(y0, wff[perm].contiguous().to(d), B=1, H=1, W=M, taps=1, bias=bff[perm].contiguous().to(d),
                       geglu=True, ln=(st, 1e-5, csf[perm].contiguous().to(d)))
    u, gate = F.linear(F.layer_norm(h, (C,), g, be, 1e-5), wg, bg).chunk(2, dim=-1)
    report("ln-folded geglu", yf, u * F.gelu(gate), rel_l2=8e-3, atol=2.5e-2)


def test_groupnorm_statistics_from_epilogue_bf16():
    from leftrefill_amd import ops, packing
    d = dev()
    N, C, H, W = 2, 320, 16, 32
    x = b16(G.T("bf.gs.x", (N, C, H, W)))
    w = b16(torch.from_numpy(weights.fill_like("bf.gs.w", (C, C, 3, 3))))
    b = torch.from_numpy(weights.fill_like("bf.gs.b", (C,)))
    y, gs = ops.gemm_conv(to_tok(x), packing.pack_conv(w, dtype=BF).to(d), B=N, H=H, W=W, taps=9,
                          bias=packing.pack_bias(b).to(d), want_gn_stats=True, splits=1)
    assert gs is not None
    g = torch.from_numpy(weights.fill_like("bf.gs.g", (C,))) + 1.0
    be = torch.from_numpy(weights.fill_like("bf.gs.be", (C,)))
    out = ops.group_norm_fused(y, N, H * W, g.to(d), be.to(d), 1e-5, True, gs)
    two_pass = ops.group_norm(y, N, H * W, g.to(d), be.to(d), 1e-5, True)
    # the epilogue takes the statistics of the fp32 accumulators, the two-pass kernel those of the bf16-rounded tensor
    report("gn fused vs two-pass", out, two_pass.float())
    report("gn fused vs fp32", from_tok(out, N, H, W), F.silu(F.group_norm(from_tok(y, N, H, W), 32, g, be, 1e-5)))


@pytest.mark.parametrize("B,heads,Nq,Nkv", [(2, 5, 128, 128), (2, 10, 64, 77), (1, 2, 200, 333), (1, 5, 2048, 2048)])
def test_attention_bf16(B, heads, Nq, Nkv):
    from leftrefill_amd import ops
    d = dev()
    C = heads * 64
    q, k, v = (b16(G.T(f"bf.att.{Nq}.{Nkv}.{n}", (B, s, C))) for n, s in (("q", Nq), ("k", Nkv), ("v", Nkv)))
    ref = unet_ref.attention(q, k, v, heads, unet_ref._Mode("fp32"))
    qd, kd, vd = (t.reshape(-1, C).to(BF).to(d) for t in (q, k, v))
    o = ops.attention(qd, kd, vd, B, heads, Nq, Nkv, 64 ** -0.5)
    assert o.dtype == BF
    # P is rounded to bf16 before the PV product: error <= 2^-8 * max|v| on top of the output rounding
    report(f"attention {Nq}x{Nkv}", o.reshape(B, Nq, C), ref, atol=1.6e-2)
    vt = ops.transpose_v(vd, B, heads, Nkv)
    o2 = ops.attention(qd, kd, vd, B, heads, Nq, Nkv, 64 ** -0.5, vt=vt)
    report(f"attention(V^T) {Nq}x{Nkv}", o2.reshape(B, Nq, C), ref, atol=1.6e-2)


def test_fused_blocks_bf16():
    """bf16 instances of the register-chained kernels of the C = 320 level (lr_xattn_block_bf16, lr_ffn_block_bf16) vs the
    oracle on bf16-rounded operands."""
    from leftrefill_amd import ops, packing
    d = dev()
    C, H, heads, B, L, Lc = 320, 1280, 5, 2, 128, 77
    m = unet_ref._Mode("fp32")
    sd = {f"a.{k}.weight": b16(torch.from_numpy(weights.fill_like(f"fb16.{k}", (C, C if k == "to_q" else 1024)))) for k in ("to_q", "to_k", "to_v")}
    sd["a.to_out.0.weight"] = b16(torch.from_numpy(weights.fill_like("fb16.to_out", (C, C))))
    sd["a.to_out.0.bias"] = torch.from_numpy(weights.fill_like("fb16.to_out.b", (C,)))
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like("fb16.g", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like("fb16.b", (C,), kind="unit"))
    x = b16(G.T("fb16.x", (B, L, C)))
    ctx = b16(G.T("fb16.ctx", (B, Lc, 1024)))
    ref = x + unet_ref.cross_attention(sd, "a", unet_ref.layer_norm(x, gamma, beta), ctx, heads, m)
    wq, bq, _ = packing.fold_layernorm(sd["a.to_q.weight"], None, gamma, beta, BF)
    xk_w, xwo = packing.pack_xattn(sd["a.to_k.weight"], sd["a.to_out.0.weight"], BF)
    ctx_t = ctx.reshape(B * Lc, -1).to(BF).to(d)
    k = ops.gemm_conv(ctx_t, xk_w.to(d), B=1, H=1, W=B * Lc, taps=1)
    v = ops.gemm_conv(ctx_t, sd["a.to_v.weight"].to(BF).to(d), B=1, H=1, W=B * Lc, taps=1)
    out = ops.xattn_block(x.reshape(B * L, C).to(BF).to(d), wq.to(d), bq.to(d), k, ops.xattn_pack_vt(v, B, heads, Lc), xwo.to(d),
                          sd["a.to_out.0.bias"].to(d), HW=L, heads=heads, Lc=Lc, eps=1e-5, scale=0.125)
    assert out.dtype == BF
    report("fused cross-attention block", out.reshape(B, L, C), ref, atol=2e-2)
    fd = {"f.net.0.proj.weight": b16(torch.from_numpy(weights.fill_like("fb16.proj", (2 * H, C)))),
          "f.net.0.proj.bias": torch.from_numpy(weights.fill_like("fb16.proj.b", (2 * H,))),
          "f.net.2.weight": b16(torch.from_numpy(weights.fill_like("fb16.ff2", (C, H)))),
          "f.net.2.bias": torch.from_numpy(weights.fill_like("fb16.ff2.b", (C,)))}
    xf = x.reshape(B * L, C)
    ref2 = xf + unet_ref.feed_forward(fd, "f", unet_ref.layer_norm(xf, gamma, beta), m)
    wf, bf, _ = packing.fold_layernorm(fd["f.net.0.proj.weight"], fd["f.net.0.proj.bias"], gamma, beta, BF)
    perm = packing.geglu_perm(H)
    out2 = ops.ffn_block(xf.to(BF).to(d), wf[perm].contiguous().to(d), bf[perm].contiguous().to(d),
                         packing.pack_pieces(fd["f.net.2.weight"], BF).to(d), fd["f.net.2.bias"].to(d), eps=1e-5)
    assert out2.dtype == BF
    report("fused feed-forward block", out2, ref2, atol=2e-2)


def test_row_resident_kernels_bf16():
    """bf16 instances of the round-6 kernels (lr_stin_block_bf16, lr_rowlin_bf16, lr_xattn_block_bf16 at C = 640) vs the oracle on
    bf16-rounded operands."""
    from leftrefill_amd import ops, packing
    d = dev()
    m = unet_ref._Mode("fp32")
    # --- stin: proj_in + LayerNorm + q|k|v at C = 320
    C, M, NQ = 320, 512, 960
    wp = b16(torch.from_numpy(weights.fill_like("rr16.wp", (C, C))))
    bp = torch.from_numpy(weights.fill_like("rr16.bp", (C,)))
    wq = b16(torch.from_numpy(weights.fill_like("rr16.wq", (NQ, C))))
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like("rr16.g", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like("rr16.b", (C,), kind="unit"))
    x = b16(G.T("rr16.x", (M, C)))
    wf, bf, _ = packing.fold_layernorm(wq, None, gamma, beta, BF)
    x1, qkv = ops.stin_block(x.to(BF).to(d), wp.to(BF).to(d), bp.to(d), wf.to(d), bf.to(d), eps=1e-5)
    assert x1.dtype == BF and qkv.dtype == BF
    report("stin x1", x1, F.linear(x, wp, bp))
    report("stin qkv", qkv, F.linear(unet_ref.layer_norm(x1.float().cpu(), gamma, beta), wq), rel_l2=8e-3, atol=2.5e-2)
    # --- rowlin at C = 640: q|k|v and GEGLU
    C, M = 640, 256
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like("rr16.g6", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like("rr16.b6", (C,), kind="unit"))
    x = b16(G.T("rr16.x6", (M, C)))
    w = b16(torch.from_numpy(weights.fill_like("rr16.w6", (1920, C))))
    wf, bf, _ = packing.fold_layernorm(w, None, gamma, beta, BF)
    out = ops.rowlin(x.to(BF).to(d), wf.to(d), bf.to(d), eps=1e-5)
    assert out.dtype == BF
    report("rowlin qkv", out, F.linear(unet_ref.layer_norm(x, gamma, beta), w), rel_l2=8e-3, atol=2.5e-2)
    H = 2560
    wg = b16(torch.from_numpy(weights.fill_like("rr16.wg", (2 * H, C))))
    bg = torch.from_numpy(weights.fill_like("rr16.bg", (2 * H,)))
    wff, bff, _ = packing.fold_layernorm(wg, bg, gamma, beta, BF)
    perm = packing.geglu_perm(H)
    og = ops.rowlin(x.to(BF).to(d), wff[perm].contiguous().to(d), bff[perm].contiguous().to(d), eps=1e-5, geglu=True)
    u, gate = F.linear(unet_ref.layer_norm(x, gamma, beta), wg, bg).chunk(2, dim=-1)
    report("rowlin geglu", og, u * F.gelu(gate), rel_l2=8e-3, atol=4e-2)
    # --- fused cross-attention block at C = 640 (with the self-attention's out-projection in front)
    heads, B, L, Lc = 10, 2, 128, 77
    sd = {f"a.{k}.weight": b16(torch.from_numpy(weights.fill_like(f"rr16.x.{k}", (C, C if k == "to_q" else 1024)))) for k in ("to_q", "to_k", "to_v")}
    sd["a.to_out.0.weight"] = b16(torch.from_numpy(weights.fill_like("rr16.x.to_out", (C, C))))
    sd["a.to_out.0.bias"] = torch.from_numpy(weights.fill_like("rr16.x.to_out.b", (C,)))
    wo1 = b16(torch.from_numpy(weights.fill_like("rr16.x.wo1", (C, C))))
    bo1 = torch.from_numpy(weights.fill_like("rr16.x.bo1", (C,)))
    xx = b16(G.T("rr16.xx", (B, L, C)))
    aa = b16(G.T("rr16.aa", (B, L, C)))
    ctx = b16(G.T("rr16.ctx", (B, Lc, 1024)))
    x1r = xx + F.linear(aa, wo1, bo1)
    ref = x1r + unet_ref.cross_attention(sd, "a", unet_ref.layer_norm(x1r, gamma, beta), ctx, heads, m)
    wqf, bq, _ = packing.fold_layernorm(sd["a.to_q.weight"], None, gamma, beta, BF)
    wq_pi = wqf[:, packing.xattn_perm(C)].contiguous()
    xk_w, xwo = packing.pack_xattn(sd["a.to_k.weight"], sd["a.to_out.0.weight"], BF)
    ctx_t = ctx.reshape(B * Lc, -1).to(BF).to(d)
    k = ops.gemm_conv(ctx_t, xk_w.to(d), B=1, H=1, W=B * Lc, taps=1)
    v = ops.gemm_conv(ctx_t, sd["a.to_v.weight"].to(BF).to(d), B=1, H=1, W=B * Lc, taps=1)
    o = ops.xattn_block(xx.reshape(B * L, C).to(BF).to(d), wq_pi.to(d), bq.to(d), k, ops.xattn_pack_vt(v, B, heads, Lc), xwo.to(d),
                        sd["a.to_out.0.bias"].to(d), HW=L, heads=heads, Lc=Lc, eps=1e-5, scale=0.125,
                        pre=(aa.reshape(B * L, C).to(BF).to(d), wo1.to(BF).to(d), bo1.to(d)))
    assert o.dtype == BF
    report("fused cross-attention block C = 640 (+ pre)", o.reshape(B, L, C), ref, rel_l2=8e-3, atol=4e-2)


def test_mv_gather_scatter_bf16():
    from leftrefill_amd import ops
    b, V, s, C = 2, 5, 4, 64
    v = V - 1
    x = b16(G.T("bf.mv.x", (b * v, 2 * s * s, C)))
    seq_ref, info = unet_ref.mv_gather(x, V, True, False)
    seq = ops.mv_gather(x.reshape(-1, C).to(BF).to(dev()), b, v, s)
    assert seq.dtype == BF and torch.equal(seq.float().cpu().reshape(seq_ref.shape), seq_ref)
    back_ref = unet_ref.mv_scatter(seq_ref, V, True, False, info)
    back = ops.mv_scatter(seq, b, v, s)
    assert torch.equal(back.float().cpu().reshape(back_ref.shape), back_ref)


def _check_grad(name, got, ref, rel_l2=1.2e-2):
    got, ref = got.float().cpu(), ref.float()
    rel = ((got - ref).norm() / ref.norm()).item()
    print(f"[bf16 bwd {name}] rel_l2 {rel:.3e} max_abs {(got - ref).abs().max().item():.3e} (|ref| max {ref.abs().max().item():.3e})")
    assert torch.isfinite(got).all() and rel <= rel_l2, (name, rel)


def test_backward_kernels_bf16():
    """Input-gradient kernels (LayerNorm, GroupNorm+SiLU, conv dgrad, GEGLU, attention) vs torch.autograd in fp32."""
    from leftrefill_amd import packing, train_ops as T
    d = dev()
    # LayerNorm
    M, C = 300, 320
    x, dy = b16(G.T("bf.lnb.x", (M, C))), b16(G.T("bf.lnb.dy", (M, C)))
    g = torch.from_numpy(weights.fill_like("bf.lnb.g", (C,))) + 1.0
    b = torch.from_numpy(weights.fill_like("bf.lnb.b", (C,)))
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), g, b, 1e-5).backward(dy)
    xd = x.to(BF).to(d).requires_grad_(True)
    T.layer_norm(xd, g.to(d), b.to(d), 1e-5).backward(dy.to(BF).to(d))
    _check_grad("layernorm", xd.grad, xr.grad)
    # GroupNorm + SiLU over a virtual concat
    N, C1, C2, H, W = 1, 640, 320, 16, 8
    x, dy = b16(G.T("bf.gnb.x", (N, C1 + C2, H, W))), b16(G.T("bf.gnb.dy", (N, C1 + C2, H, W)))
    g = torch.from_numpy(weights.fill_like("bf.gnb.g", (C1 + C2,))) + 1.0
    b = torch.from_numpy(weights.fill_like("bf.gnb.b", (C1 + C2,)))
    xr = x.clone().requires_grad_(True)
    F.silu(F.group_norm(xr, 32, g, b, 1e-5)).backward(dy)
    x1, x2 = to_tok(x[:, :C1]).requires_grad_(True), to_tok(x[:, C1:]).requires_grad_(True)
    T.group_norm(x1, N, H * W, g.to(d), b.to(d), 1e-5, True, x2).backward(to_tok(dy))
    _check_grad("groupnorm dx1", from_tok(x1.grad, N, H, W), xr.grad[:, :C1])
    _check_grad("groupnorm dx2", from_tok(x2.grad, N, H, W), xr.grad[:, C1:])
    # conv 3x3 (+ stride-2, + nearest-up) dgrad
    for name, kw, (Hs, Ws, Ho, Wo) in (("c3", {}, (12, 20, 12, 20)), ("s2", {"stride": 2}, (12, 20, 6, 10)),
                                       ("up", {"up": 1}, (6, 10, 12, 20))):
        x = b16(G.T(f"bf.cb.{name}.x", (2, 320, Hs, Ws)))
        w = b16(torch.from_numpy(weights.fill_like(f"bf.cb.{name}.w", (320, 320, 3, 3))))
        dy = b16(G.T(f"bf.cb.{name}.dy", (2, 320, Ho, Wo)))
        xr = x.clone().requires_grad_(True)
        xin = F.interpolate(xr, scale_factor=2, mode="nearest") if kw.get("up") else xr
        F.conv2d(xin, w, None, stride=kw.get("stride", 1), padding=1).backward(dy)
        xd = to_tok(x).requires_grad_(True)
        T.gemm_conv(xd, packing.pack_conv(w, dtype=BF).to(d), B=2, H=Ho, W=Wo, Hs=Hs, Ws=Ws, taps=9, **kw).backward(to_tok(dy))
        _check_grad(f"conv {name}", from_tok(xd.grad, 2, Hs, Ws), xr.grad)
    # GEGLU
    C, M = 320, 200
    x, dy = b16(G.T("bf.ggb.x", (M, C))), b16(G.T("bf.ggb.dy", (M, 4 * C)))
    w = b16(torch.from_numpy(weights.fill_like("bf.ggb.w", (8 * C, C))))
    b = torch.from_numpy(weights.fill_like("bf.ggb.b", (8 * C,)))
    xr = x.clone().requires_grad_(True)
    u, gate = F.linear(xr, w, b).chunk(2, dim=-1)
    (u * F.gelu(gate)).backward(dy)
    wp, bp = packing.pack_geglu(w, b, dtype=BF)
    xd = x.to(BF).to(d).requires_grad_(True)
    T.gemm_conv(xd, wp.to(d), B=1, H=1, W=M, taps=1, bias=bp.to(d), geglu=True).backward(dy.to(BF).to(d))
    _check_grad("geglu", xd.grad, xr.grad)
    # attention (self, fused QKV layout; cross with Nkv = 77)
    for B, heads, Nq, Nkv in ((2, 5, 256, 256), (2, 5, 192, 77)):
        Cc = heads * 64
        q, k, v = (b16(G.T(f"bf.ab.{Nq}.{Nkv}.{n}", (B, s, Cc))) for n, s in (("q", Nq), ("k", Nkv), ("v", Nkv)))
        do = b16(G.T(f"bf.ab.{Nq}.{Nkv}.do", (B, Nq, Cc)))
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
        unet_ref.attention(qr, kr, vr, heads, unet_ref._Mode("fp32")).backward(do)
        qd, kd, vd = (t.reshape(-1, Cc).to(BF).to(d).requires_grad_(True) for t in (q, k, v))
        T.attention(qd, kd, vd, B, heads, Nq, Nkv, 64 ** -0.5).backward(do.reshape(-1, Cc).to(BF).to(d))
        for n, a_, r_ in (("dq", qd, qr), ("dk", kd, kr), ("dv", vd, vr)):
            _check_grad(f"attention {Nq}x{Nkv} {n}", a_.grad.reshape(r_.shape), r_.grad, rel_l2=1.5e-2)


def test_unet_forward_bf16_vs_oracle():
    """Whole UNet (the golden trajectory config and the shipped width) with compute_dtype = bfloat16 against the fp32 oracle.
    Also prints the oracle's own bf16-autocast emulation error as the yardstick the reference itself would achieve."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    for name, N, H, W, bound in ((G.TRAJ_CONFIG, 2, 32, 64, 2.5e-2), ("FULL", 2, 16, 32, 2.5e-2)):
        cfg = G.CONFIGS[name]
        sd = G.unet_state(name)
        m = UNetModel(**cfg.kwargs())
        m.load_state_dict(sd, strict=True)
        m = m.to(dev()).eval()
        m.compute_dtype = BF
        x, t, ctx = G.unet_inputs("bf16_" + name, cfg, N, H, W, [981, 1])
        ref = unet_ref.unet_forward(sd, cfg, x, t, ctx)
        with torch.no_grad():
            out = m(x.to(dev()), t.to(dev()), context=ctx.to(dev()))
            eager = None
            m.use_hip_graph = False
            eager = m(x.to(dev()), t.to(dev()), context=ctx.to(dev()))
        assert out.dtype == BF and torch.equal(out, eager)          # captured graph == eager launches, bit for bit
        o = out.float().cpu()
        rel = ((o - ref).norm() / ref.norm()).item()
        print(f"[bf16 unet {name}] rel_l2 {rel:.3e} max_abs {(o - ref).abs().max().item():.3e} (|ref| max {ref.abs().max().item():.3e})")
        assert torch.isfinite(o).all() and rel <= bound
        # switching back re-packs in fp16 and reproduces the fp16 result exactly
        m.compute_dtype = torch.float16
        with torch.no_grad():
            o16 = m(x.to(dev()), t.to(dev()), context=ctx.to(dev()))
        assert o16.dtype == torch.float16
        rel16 = ((o16.float().cpu() - ref).norm() / ref.norm()).item()
        print(f"[bf16 unet {name}] fp16 re-pack rel_l2 {rel16:.3e}")
        assert rel16 < rel


@pytest.mark.parametrize("case,B,h,w,ts", G.TRAIN_CASES, ids=[c[0] for c in G.TRAIN_CASES])
def test_training_step_bf16_vs_reference_golden(golden, case, B, h, w, ts):
    """configs[4] in bf16: RefInpaintLDM.p_losses on the HIP path and its backward to the context, WITHOUT loss scaling
    (bf16 has fp32's exponent range), against the loss / gradient the real reference produced in fp32 on CPU
    (tests/golden/train.npz)."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM
    g = golden("train")
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    m = RefInpaintLDM(first_stage_config={"target": "torch.nn.Identity"}, cond_stage_config={"target": "torch.nn.Identity"},
                      unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": cfg.kwargs()},
                      conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                      timesteps=1000, channels=4, data_config={"img_size": 256})
    m.model.diffusion_model.load_state_dict(G.unet_state(G.TRAJ_CONFIG), strict=True)
    m = m.to(dev()).train()
    m.model.diffusion_model.compute_dtype = BF
    for p in m.parameters():
        p.requires_grad_(False)
    x_start = G.T(case + ".x_start", (B, 4, h, w)).to(dev())
    noise = G.T(case + ".noise", (B, 4, h, w)).to(dev())
    c_concat = G.T(case + ".c_concat", (B, 5, h, w)).to(dev())
    c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim)).to(dev()).requires_grad_(True)
    t = torch.tensor(ts, dtype=torch.long, device=dev())
    loss, ld = m.p_losses(x_start, {"c_concat": [c_concat], "c_crossattn": [c_cross]}, t, noise=noise)
    loss.backward()
    grad = c_cross.grad.float().cpu()
    ref = torch.from_numpy(g[case + ".dctx"])
    rel = ((grad - ref).norm() / ref.norm()).item()
    print(f"[bf16 train {case}] loss {loss.item():.6f} (reference {float(g[case + '.loss']):.6f}); d/dcontext rel_l2 {rel:.3e}, "
          f"|grad| max {ref.abs().max().item():.3e}")
    assert abs(loss.item() - float(g[case + ".loss"])) <= 1e-2 * float(g[case + ".loss"])
    assert torch.isfinite(grad).all() and rel <= 6e-2
    # gradients of realistic size (dLoss/deps ~ 1e-6 at B = 16, 64 x 128 x 4) must not take an fp16 detour: the
    # same backward with the loss scaled by 2^-14 has to give the same gradient up to that factor (bf16 keeps fp32's exponent)
    c_cross.grad = None
    loss2, _ = m.p_losses(x_start, {"c_concat": [c_concat], "c_crossattn": [c_cross]}, t, noise=noise)
    (loss2 * 2.0 ** -14).backward()
    small = c_cross.grad.float().cpu() * 2.0 ** 14
    rel2 = ((small - grad).norm() / grad.norm()).item()
    print(f"[bf16 train {case}] gradient from a 2^-14-scaled loss: rel_l2 {rel2:.3e} vs the unscaled one")
    assert rel2 <= 1e-2
