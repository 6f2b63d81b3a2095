"""GPU parity of the backward (input-gradient) kernels against torch.autograd on the fp32 reference formulation of each
operator (training with frozen weights).  Inputs / weights / upstream gradients are rounded
to fp16 first so both sides differentiate the same function; the HIP side keeps fp16 activations and gradients with
fp32 accumulation, so the bound is the north-star rtol 2e-3 / atol 1e-3 scaled by the gradient's magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, weights  # noqa: E402


def dev():
    return torch.device("cuda:0")


def h16(t):
    return t.half().float()


def to_tok(x):       # NCHW fp32 -> token-major fp16 on the device
    n, c, hh, ww = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * hh * ww, c).half().contiguous().to(dev())


def from_tok(t, n, hh, ww):
    return t.float().cpu().reshape(n, hh, ww, -1).permute(0, 3, 1, 2)


def check(name, got, ref, rtol=2e-3, atol_scale=2e-3):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all(), name
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"[bwd {name}] rel_l2 {rel:.3e} max_abs {err.max().item():.3e} (|ref| max {scale:.3e})")
    assert rel < 3e-3, (name, rel)
    assert (err <= atol_scale * scale + rtol * ref.abs()).all(), name


@pytest.mark.parametrize("M,C", [(300, 320), (77, 640), (130, 1280)])
def test_layernorm_backward(M, C):
    from leftrefill_amd import train_ops as T
    x = h16(G.T(f"lnb.x{C}", (M, C)))
    dy = h16(G.T(f"lnb.dy{C}", (M, C)))
    g = torch.from_numpy(weights.fill_like(f"lnb.{C}.weight", (C,)))
    b = torch.from_numpy(weights.fill_like(f"lnb.{C}.bias", (C,)))
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), g, b, 1e-5).backward(dy)
    xd = x.half().to(dev()).requires_grad_(True)
    y = T.layer_norm(xd, g.to(dev()), b.to(dev()), 1e-5)
    y.backward(dy.half().to(dev()))
    check(f"layernorm {M}x{C}", xd.grad, xr.grad)


@pytest.mark.parametrize("N,C1,C2,H,W,silu,eps", [(2, 320, 0, 8, 16, True, 1e-5), (1, 640, 320, 16, 8, True, 1e-5),
                                                   (2, 320, 0, 16, 16, False, 1e-6), (1, 1280, 1280, 8, 8, True, 1e-5)])
def test_groupnorm_backward(N, C1, C2, H, W, silu, eps):
    from leftrefill_amd import train_ops as T
    C = C1 + C2
    tag = f"gnb.{C1}.{C2}.{H}"
    x = h16(G.T(tag + ".x", (N, C, H, W)))
    dy = h16(G.T(tag + ".dy", (N, C, H, W)))
    g = torch.from_numpy(weights.fill_like(tag + ".weight", (C,)))
    b = torch.from_numpy(weights.fill_like(tag + ".bias", (C,)))
    xr = x.clone().requires_grad_(True)
    y = F.group_norm(xr, 32, g, b, eps)
    (F.silu(y) if silu else y).backward(dy)
    x1 = to_tok(x[:, :C1]).requires_grad_(True)
    x2 = to_tok(x[:, C1:]).requires_grad_(True) if C2 else None
    out = T.group_norm(x1, N, H * W, g.to(dev()), b.to(dev()), eps, silu, x2)
    out.backward(to_tok(dy))
    check(f"groupnorm {tag} dx1", from_tok(x1.grad, N, H, W), xr.grad[:, :C1])
    if C2:
        check(f"groupnorm {tag} dx2", from_tok(x2.grad, N, H, W), xr.grad[:, C1:])


@pytest.mark.parametrize("M,C", [(300, 320), (130, 1280)])
def test_layernorm_fork_backward(M, C):
    """x + f(LayerNorm(x)) with the residual taken from the fork's second output: ONE backward kernel adds the residual branch's
    gradient in fp32 (lr_layernorm_bwd_res), the result is the autograd gradient of the same expression."""
    from leftrefill_amd import train_ops as T
    x = h16(G.T(f"lnf.x{C}", (M, C)))
    dy = h16(G.T(f"lnf.dy{C}", (M, C)))
    g = torch.from_numpy(weights.fill_like(f"lnf.{C}.weight", (C,)))
    b = torch.from_numpy(weights.fill_like(f"lnf.{C}.bias", (C,)))
    xr = x.clone().requires_grad_(True)
    (2.0 * F.layer_norm(xr, (C,), g, b, 1e-5) + xr).backward(dy)
    xd = x.half().to(dev()).requires_grad_(True)
    n, xa = T.layer_norm_fork(xd, g.to(dev()), b.to(dev()), 1e-5)
    (2.0 * n + xa).backward(dy.half().to(dev()))
    check(f"layernorm fork {M}x{C}", xd.grad, xr.grad)
    # the fork's second output alone (the normalised branch unused) passes its gradient through
    xd2 = x.half().to(dev()).requires_grad_(True)
    _, xa2 = T.layer_norm_fork(xd2, g.to(dev()), b.to(dev()), 1e-5)
    xa2.backward(dy.half().to(dev()))
    assert torch.equal(xd2.grad, dy.half().to(dev()))


@pytest.mark.parametrize("N,C1,C2,H,W,silu,eps", [(2, 320, 0, 8, 16, True, 1e-5), (1, 640, 320, 16, 8, True, 1e-5),
                                                   (2, 320, 0, 16, 16, False, 1e-6)])
def test_groupnorm_fork_backward(N, C1, C2, H, W, silu, eps):
    """f(GroupNorm([x1 | x2])) + [x1 | x2] through the fork: the residual gradients of both sources join inside lr_groupnorm_bwd_res."""
    from leftrefill_amd import train_ops as T
    C = C1 + C2
    tag = f"gnf.{C1}.{C2}.{H}"
    x = h16(G.T(tag + ".x", (N, C, H, W)))
    dy = h16(G.T(tag + ".dy", (N, C, H, W)))
    g = torch.from_numpy(weights.fill_like(tag + ".weight", (C,)))
    b = torch.from_numpy(weights.fill_like(tag + ".bias", (C,)))
    xr = x.clone().requires_grad_(True)
    y = F.group_norm(xr, 32, g, b, eps)
    ((F.silu(y) if silu else y) + 0.5 * xr).backward(dy)
    x1 = to_tok(x[:, :C1]).requires_grad_(True)
    x2 = to_tok(x[:, C1:]).requires_grad_(True) if C2 else None
    out, a1, a2 = T.group_norm_fork(x1, N, H * W, g.to(dev()), b.to(dev()), eps, silu, x2)
    res = 0.5 * (a1 if a2 is None else torch.cat([a1, a2], dim=1))
    (out + res).backward(to_tok(dy))
    check(f"groupnorm fork {tag} dx1", from_tok(x1.grad, N, H, W), xr.grad[:, :C1])
    if C2:
        check(f"groupnorm fork {tag} dx2", from_tok(x2.grad, N, H, W), xr.grad[:, C1:])


@pytest.mark.parametrize("N,Cin,C,H,W,cat", [(2, 320, 320, 16, 16, False), (1, 640, 640, 32, 16, False), (2, 320, 320, 16, 16, True)])
def test_groupnorm_backward_with_producer_statistics(N, Cin, C, H, W, cat):
    """The differentiable forward takes the GroupNorm statistics out of the producing conv's epilogue (per-group sums, or per-channel
    partials + finalize for a virtual concat): conv -> GroupNorm + SiLU differentiated against torch.autograd on the same function."""
    from leftrefill_amd import engine as E, packing, train_ops as T
    tag = f"gnp.{Cin}.{C}.{H}.{int(cat)}"
    x = h16(G.T(tag + ".x", (N, Cin, H, W)))
    w = h16(torch.from_numpy(weights.fill_like(tag + ".w", (C, Cin, 3, 3))))
    bb = torch.from_numpy(weights.fill_like(tag + ".b", (C,)))
    Cn = 2 * C if cat else C
    g = torch.from_numpy(weights.fill_like(tag + ".weight", (Cn,)))
    b = torch.from_numpy(weights.fill_like(tag + ".bias", (Cn,)))
    dy = h16(G.T(tag + ".dy", (N, Cn, H, W)))
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, bb, padding=1)
    yr = torch.cat([yr, 0.5 * yr], 1) if cat else yr
    F.silu(F.group_norm(h16(yr.detach()) + (yr - yr.detach()), 32, g, b, 1e-5)).backward(dy)      # (the HIP side normalises the rounded conv output)
    wp = packing.pack_conv(w, cin_pad=Cin).to(dev())
    bp = packing.pack_bias(bb).to(dev())
    x1 = to_tok(x).requires_grad_(True)
    y1, gs1 = T.gemm_conv(x1, wp, B=N, H=H, W=W, taps=9, bias=bp, want_gn_stats=True)
    assert gs1 is not None and y1.requires_grad and not gs1[0].requires_grad
    if cat:
        y2, gs2 = T.gemm_conv(x1, (0.5 * wp.float()).half(), B=N, H=H, W=W, taps=9, bias=0.5 * bp, want_gn_stats=True)
        act = E.Act(y1, N, H, W, tok2=y2, gs=gs1, gs2=gs2)
    else:
        act = E.Act(y1, N, H, W, gs=gs1)
    st = E._gn_train_stats(act)
    assert st is not None and st[0] == ("channels" if cat else "groups")
    pn = type("PN", (), {"g": g.to(dev()), "b": b.to(dev()), "eps": 1e-5})()
    out = E.gn(act, pn, True)      # (goes through train_ops: the inputs require grad)
    out.tok.backward(to_tok(dy))
    check(f"conv -> groupnorm (producer statistics) {tag} dx", from_tok(x1.grad, N, H, W), xr.grad, rtol=4e-3, atol_scale=4e-3)


def _conv_bwd_case(name, N, Cin, Cout, H, W, taps=9, stride=1, up=0, C2=0, resid=False):
    from leftrefill_amd import packing, train_ops as T
    Ct = Cin + C2
    Hs, Ws = H, W
    if stride == 2:
        Hs, Ws = 2 * H, 2 * W
    if up:
        Hs, Ws = H // 2, W // 2
    k = 3 if taps == 9 else 1
    x = h16(G.T(name + ".x", (N, Ct, Hs, Ws)))
    w = h16(torch.from_numpy(weights.fill_like(name + ".w", (Cout, Ct, k, k))))
    b = torch.from_numpy(weights.fill_like(name + ".b", (Cout,)))
    dy = h16(G.T(name + ".dy", (N, Cout, H, W)))
    xr = x.clone().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    y = F.conv2d(xin, w, b, stride=stride, padding=1 if taps == 9 else 0)
    rs = None
    if resid:
        rs = h16(G.T(name + ".rs", (N, Cout, H, W))).requires_grad_(True)
        y = y + rs
    y.backward(dy)
    wp = packing.pack_conv(w, cin_pad=Ct).to(dev())
    bp = packing.pack_bias(b).to(dev())
    x1 = to_tok(x[:, :Cin]).requires_grad_(True)
    x2 = to_tok(x[:, Cin:]).requires_grad_(True) if C2 else None
    rd = to_tok(rs.detach()).requires_grad_(True) if resid else None
    out = T.gemm_conv(x1, wp, B=N, H=H, W=W, Hs=Hs, Ws=Ws, taps=taps, stride=stride, up=up, x2=x2, bias=bp, resid=rd)
    out.backward(to_tok(dy))
    check(name + " dx1", from_tok(x1.grad, N, Hs, Ws), xr.grad[:, :Cin])
    if C2:
        check(name + " dx2", from_tok(x2.grad, N, Hs, Ws), xr.grad[:, Cin:])
    if resid:
        assert torch.equal(rd.grad, to_tok(dy))


def test_conv_linear_input_gradients():
    _conv_bwd_case("cb_lin", 1, 320, 640, 10, 30, taps=1)
    _conv_bwd_case("cb_lin_res", 2, 640, 640, 8, 8, taps=1, resid=True)
    _conv_bwd_case("cb_c3", 2, 320, 320, 12, 20)
    _conv_bwd_case("cb_c3_cat", 1, 640, 320, 8, 16, C2=320, resid=True)
    _conv_bwd_case("cb_c1_cat", 2, 640, 320, 8, 8, taps=1, C2=320)
    _conv_bwd_case("cb_s2", 2, 320, 320, 6, 10, stride=2)
    _conv_bwd_case("cb_up", 1, 640, 640, 8, 12, up=1)
    _conv_bwd_case("cb_out", 2, 320, 64, 8, 16)          # padded output conv (4 -> 64 channels)


@pytest.mark.parametrize("N,Ch,C1,C2,Cout,H,W", [(2, 640, 320, 0, 640, 16, 16), (1, 320, 640, 320, 320, 16, 32)])
def test_skip_extended_conv_backward(N, Ch, C1, C2, Cout, H, W):
    """conv3x3(h) + skip_connection([s1 | s2]) as ONE launch in the differentiable forward (lr_gemm_args.skip1); the backward multiplies dY by
    the two layers' own weights: dh, ds1, ds2 against torch.autograd of the two-conv formulation."""
    from leftrefill_amd import packing, train_ops as T
    tag = f"csk.{Ch}.{C1}.{C2}.{Cout}"
    h = h16(G.T(tag + ".h", (N, Ch, H, W)))
    sx = h16(G.T(tag + ".s", (N, C1 + C2, H, W)))
    w3 = h16(torch.from_numpy(weights.fill_like(tag + ".w3", (Cout, Ch, 3, 3))))
    w1 = h16(torch.from_numpy(weights.fill_like(tag + ".w1", (Cout, C1 + C2, 1, 1))))
    b3 = torch.from_numpy(weights.fill_like(tag + ".b3", (Cout,)))
    b1 = torch.from_numpy(weights.fill_like(tag + ".b1", (Cout,)))
    dy = h16(G.T(tag + ".dy", (N, Cout, H, W)))
    hr, sr = h.clone().requires_grad_(True), sx.clone().requires_grad_(True)
    yr = F.conv2d(hr, w3, b3, padding=1) + F.conv2d(sr, w1, b1)
    yr.backward(dy)
    wp3 = packing.pack_conv(w3, cin_pad=Ch).to(dev())
    wp1 = packing.pack_conv(w1, cin_pad=C1 + C2).to(dev())
    wf = torch.cat([wp3, wp1], dim=1).contiguous()
    bf = (packing.pack_bias(b3) + packing.pack_bias(b1)).to(dev())
    hd = to_tok(h).requires_grad_(True)
    s1 = to_tok(sx[:, :C1]).requires_grad_(True)
    s2 = to_tok(sx[:, C1:]).requires_grad_(True) if C2 else None
    y, gs = T.gemm_conv(hd, wf, B=N, H=H, W=W, taps=9, bias=bf, skip=(s1, s2), skip_parts=(wp3, wp1), want_gn_stats=True)
    check(tag + " forward", from_tok(y.detach(), N, H, W), yr.detach(), rtol=3e-3, atol_scale=3e-3)
    assert gs is not None and not gs[0].requires_grad
    y.backward(to_tok(dy))
    check(tag + " dh", from_tok(hd.grad, N, H, W), hr.grad)
    check(tag + " ds1", from_tok(s1.grad, N, H, W), sr.grad[:, :C1])
    if C2:
        check(tag + " ds2", from_tok(s2.grad, N, H, W), sr.grad[:, C1:])


def test_geglu_backward():
    from leftrefill_amd import packing, train_ops as T
    C, M = 320, 200
    x = h16(G.T("ggb.x", (M, C)))
    w = h16(torch.from_numpy(weights.fill_like("ggb.w", (8 * C, C))))
    b = torch.from_numpy(weights.fill_like("ggb.b", (8 * C,)))
    dy = h16(G.T("ggb.dy", (M, 4 * C)))
    xr = x.clone().requires_grad_(True)
    u, gate = F.linear(xr, w, b).chunk(2, dim=-1)
    (u * F.gelu(gate)).backward(dy)
    wp, bp = packing.pack_geglu(w, b)
    xd = x.half().to(dev()).requires_grad_(True)
    y = T.gemm_conv(xd, wp.to(dev()), B=1, H=1, W=M, taps=1, bias=bp.to(dev()), geglu=True)
    y.backward(dy.half().to(dev()))
    check("geglu dx", xd.grad, xr.grad)


@pytest.mark.parametrize("B,heads,Nq,Nkv", [(1, 1, 64, 64), (2, 2, 128, 128), (1, 3, 300, 200), (2, 5, 256, 77), (1, 2, 512, 1024),
                                            (1, 2, 2048, 2048), (2, 5, 2048, 77), (1, 2, 1000, 200)])      # (the last two: query-split dK / dV)
def test_attention_backward(B, heads, Nq, Nkv):
    """dQ, dK, dV of softmax(q k^T / 8) v per head vs torch.autograd on the fp32 formulation; q / k / v are strided column
    slices of one fused buffer like in the UNet; the forward output of the lse-saving kernel must equal the plain one."""
    from leftrefill_amd import ops, train_ops as T
    d = dev()
    C = heads * 64
    tag = f"attb.{Nq}.{Nkv}.{heads}"
    q = h16(G.T(tag + ".q", (B, Nq, C)))
    k = h16(G.T(tag + ".k", (B, Nkv, C)))
    v = h16(G.T(tag + ".v", (B, Nkv, C)))
    do = h16(G.T(tag + ".do", (B, Nq, C)))
    qr, kr, vr = (t_.clone().requires_grad_(True) for t_ in (q, k, v))

    def split(t_, n):
        return t_.reshape(B, n, heads, 64).permute(0, 2, 1, 3)

    o_ref = F.scaled_dot_product_attention(split(qr, Nq), split(kr, Nkv), split(vr, Nkv), scale=0.125)
    o_ref = o_ref.permute(0, 2, 1, 3).reshape(B, Nq, C)
    o_ref.backward(do)
    qd = q.reshape(B * Nq, C).half().to(d).requires_grad_(True)
    kv = torch.cat([k, v], -1).reshape(B * Nkv, 2 * C).half().to(d).requires_grad_(True)
    o = T.attention(qd, kv[:, :C], kv[:, C:], B, heads, Nq, Nkv, 0.125)
    with torch.no_grad():
        o_plain = ops.attention(qd.detach(), kv.detach()[:, :C], kv.detach()[:, C:], B, heads, Nq, Nkv, 0.125)
    # the training forward keeps the scale outside the QK^T operands (the backward recomputes P from the unscaled q, k and the saved
    # log-sum-exp); the inference kernel folds scale * log2(e) into Q: same result to fp16 rounding, not bit for bit
    assert (o.detach().float() - o_plain.float()).abs().max().item() <= 2e-3
    o.backward(do.reshape(B * Nq, C).half().to(d))
    check(tag + " dq", qd.grad.reshape(B, Nq, C), qr.grad)
    check(tag + " dk", kv.grad[:, :C].reshape(B, Nkv, C), kr.grad)
    check(tag + " dv", kv.grad[:, C:].reshape(B, Nkv, C), vr.grad)


def test_unet_context_gradient_matches_oracle_autograd():
    """Whole-UNet training step on the HIP path (frozen weights): d loss / d context through every block -- conv / linear
    dgrad, GroupNorm / LayerNorm / GEGLU / attention backward, stride-2 and nearest-up convs, skip concats -- against
    torch.autograd on the CPU oracle (fp32).  The bound is the fp16 noise of the chain, measured with the oracle's own
    fp16-autocast emulation differentiated the same way."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import unet_ref
    cfg = G.CONFIGS["MID"]
    sd = G.unet_state("MID")
    m = UNetModel(**cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).eval()
    N, H, W = 2, 16, 32
    x, t, ctx = G.unet_inputs("bwd_mid", cfg, N, H, W, [501, 21])
    deps = h16(G.T("bwd_mid.deps", (N, 4, H, W)))

    def oracle_grad(mode):
        c = ctx.clone().requires_grad_(True)
        out = unet_ref.unet_forward.__wrapped__(sd, cfg, x, t, c, mode=mode)     # the undecorated (grad-enabled) function
        out.backward(deps)
        return out.detach(), c.grad

    out_ref, g_ref = oracle_grad("fp32")
    _, g_emul = oracle_grad("autocast16")
    cd = ctx.to(dev()).requires_grad_(True)
    out = m(x.to(dev()), t.to(dev()), context=cd)
    out.float().backward(deps.to(dev()))
    g = cd.grad.float().cpu()
    assert torch.isfinite(g).all()
    rel_out = ((out.float().cpu() - out_ref).norm() / out_ref.norm()).item()
    rel = ((g - g_ref).norm() / g_ref.norm()).item()
    rel_e = ((g_emul - g_ref).norm() / g_ref.norm()).item()
    print(f"[bwd unet MID] forward rel_l2 {rel_out:.3e}; d/dcontext rel_l2 {rel:.3e} (autocast16 emulation {rel_e:.3e}), "
          f"|grad| max {g_ref.abs().max().item():.3e}")
    assert rel <= max(2.0 * rel_e, 1e-2), (rel, rel_e)
    # a second backward through a fresh forward is bit-identical (no atomics anywhere in the backward kernels)
    cd2 = ctx.to(dev()).requires_grad_(True)
    m(x.to(dev()), t.to(dev()), context=cd2).float().backward(deps.to(dev()))
    assert torch.equal(cd2.grad, cd.grad)
    # the reference's activation checkpointing (use_checkpoint: recompute every block in the backward) gives the same bits
    m.recompute_in_backward = True
    cd3 = ctx.to(dev()).requires_grad_(True)
    m(x.to(dev()), t.to(dev()), context=cd3).float().backward(deps.to(dev()))
    m.recompute_in_backward = False
    assert torch.equal(cd3.grad, cd.grad)


@pytest.mark.parametrize("case,B,h,w,ts", G.TRAIN_CASES, ids=[c[0] for c in G.TRAIN_CASES])
def test_training_step_vs_reference_golden(golden, case, B, h, w, ts):
    """RefInpaintLDM.p_losses on the HIP path (q_sample -> apply_model -> UNet -> MSE) and its backward to the context,
    against the loss / gradient the REAL reference produced on CPU (tests/golden/train.npz).  fp16 gradients need the
    loss scaling the reference's `--fp16` trainer applies (Lightning native AMP): the loss is scaled by 2^14 here."""
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
    for p in m.parameters():
        p.requires_grad_(False)          # frozen backbone: result-identical to the reference, whose optimizer never steps them
    x_start = G.T(case + ".x_start", (B, 4, h, w)).to(dev())
    noise = G.T(case + ".noise", (B, 4, h, w)).to(dev())
    c_concat = G.T(case + ".c_concat", (B, 5, h, w)).to(dev())
    c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim)).to(dev()).requires_grad_(True)
    t = torch.tensor(ts, dtype=torch.long, device=dev())
    loss, ld = m.p_losses(x_start, {"c_concat": [c_concat], "c_crossattn": [c_cross]}, t, noise=noise)
    scale = 2.0 ** 14
    (loss * scale).backward()
    grad = c_cross.grad.float().cpu() / scale
    ref = torch.from_numpy(g[case + ".dctx"])
    rel = ((grad - ref).norm() / ref.norm()).item()
    print(f"[bwd train {case}] loss {loss.item():.6f} (reference {float(g[case + '.loss']):.6f}); d/dcontext rel_l2 {rel:.3e}, "
          f"|grad| max {ref.abs().max().item():.3e}")
    assert set(ld) == {"train/loss_simple", "train/loss_vlb", "train/loss"}
    assert abs(loss.item() - float(g[case + ".loss"])) <= 2e-3 * float(g[case + ".loss"])
    assert abs(ld["train/loss_vlb"].item() - float(g[case + ".loss_vlb"])) <= 3e-3 * abs(float(g[case + ".loss_vlb"]))
    assert torch.isfinite(grad).all() and rel <= 1e-2


def test_unet_full_width_context_gradient():
    """The shipped 866 M-parameter width (latent 16x32, N=2): d loss / d context on the HIP path vs torch.autograd on the CPU
    oracle (pinned to the reference at this width by goldens G4)."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from oracle import unet_ref
    cfg = G.CONFIGS["FULL"]
    sd = G.unet_state("FULL")
    m = UNetModel(**cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).eval()
    N, H, W = 2, 16, 32
    x, t, ctx = G.unet_inputs("bwd_full", cfg, N, H, W, [981, 1])
    deps = h16(G.T("bwd_full.deps", (N, 4, H, W)))
    c = ctx.clone().requires_grad_(True)
    unet_ref.unet_forward.__wrapped__(sd, cfg, x, t, c).backward(deps)
    cd = ctx.to(dev()).requires_grad_(True)
    m(x.to(dev()), t.to(dev()), context=cd).float().backward(deps.to(dev()))
    g, g_ref = cd.grad.float().cpu(), c.grad
    rel = ((g - g_ref).norm() / g_ref.norm()).item()
    print(f"[bwd unet FULL] d/dcontext rel_l2 {rel:.3e}, |grad| max {g_ref.abs().max().item():.3e}")
    assert torch.isfinite(g).all() and rel <= 1e-2


@pytest.mark.parametrize("V,concat,b,H,W", [(3, True, 1, 8, 16), (2, False, 2, 8, 8)], ids=["v3_concat_target", "v2_plain"])
def test_multiview_unet_context_gradient(V, concat, b, H, W):
    """Multi-view UNet (re-arranged cross-view self-attention): d loss / d context incl. the gather / scatter backward
    (the shared target slot sums the gradients of every canvas' right half) vs torch.autograd on the CPU oracle."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from ldm.modules.diffusionmodules.multiview_unet import MultiViewUnetModel
    from oracle import unet_ref
    cfg = G.mv_config(V, concat)
    sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
    m = MultiViewUnetModel(**cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).eval()
    n = b * (V - 1 if concat else V)
    x, t, ctx = G.unet_inputs(f"bwd_mv{V}", cfg, n, H, W, [501] * n)
    deps = h16(G.T(f"bwd_mv{V}.deps", (n, 4, H, W)))
    c = ctx.clone().requires_grad_(True)
    unet_ref.unet_forward.__wrapped__(sd, cfg, x, t, c).backward(deps)
    cd = ctx.to(dev()).requires_grad_(True)
    m(x.to(dev()), t.to(dev()), context=cd).float().backward(deps.to(dev()))
    g, g_ref = cd.grad.float().cpu(), c.grad
    rel = ((g - g_ref).norm() / g_ref.norm()).item()
    print(f"[bwd unet MV V={V} concat={concat}] d/dcontext rel_l2 {rel:.3e}, |grad| max {g_ref.abs().max().item():.3e}")
    assert torch.isfinite(g).all() and rel <= 1e-2


def test_attention_backward_with_late_score_spike():
    """A key that matches one query strongly late in the sequence: the forward's deferred running-max rescale and the saved
    log-sum-exp must still give exact probabilities in the backward."""
    from leftrefill_amd import train_ops as T
    d = dev()
    N = 512
    q = h16(G.T("attbs.q", (1, N, 64)))
    k = h16(G.T("attbs.k", (1, N, 64)))
    v = h16(G.T("attbs.v", (1, N, 64)))
    do = h16(G.T("attbs.do", (1, N, 64)))
    k[0, 450] = q[0, 7] * 6.0
    k[0, 70] = q[0, 300] * 4.0
    qr, kr, vr = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
    F.scaled_dot_product_attention(qr[:, None], kr[:, None], vr[:, None], scale=0.125)[:, 0].backward(do)
    qd, kd, vd = (t_.reshape(N, 64).half().to(d).requires_grad_(True) for t_ in (q, k, v))
    T.attention(qd, kd, vd, 1, 1, N, N, 0.125).backward(do.reshape(N, 64).half().to(d))
    check("spike dq", qd.grad.reshape(1, N, 64), qr.grad)
    check("spike dk", kd.grad.reshape(1, N, 64), kr.grad)
    check("spike dv", vd.grad.reshape(1, N, 64), vr.grad)
