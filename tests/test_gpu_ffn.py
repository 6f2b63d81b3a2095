"""GPU parity of the fused feed-forward block (lr_ffn_block_f16) against the CPU oracle.

Reference semantics: `x = self.ff(self.norm3(x)) + x` (ldm/modules/attention.py:282), FeedForward with GEGLU (attention.py:51-78);
oracle: unet_ref.layer_norm + unet_ref.feed_forward.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402
from tests.test_gpu_ops import dev, h16, report  # noqa: E402

C = 320


def _params(tag, H):
    sd = {"f.net.0.proj.weight": h16(torch.from_numpy(weights.fill_like(f"ffb.{tag}.proj.weight", (2 * H, C)))),
          "f.net.0.proj.bias": torch.from_numpy(weights.fill_like(f"ffb.{tag}.proj.bias", (2 * H,))),
          "f.net.2.weight": h16(torch.from_numpy(weights.fill_like(f"ffb.{tag}.net2.weight", (C, H)))),
          "f.net.2.bias": torch.from_numpy(weights.fill_like(f"ffb.{tag}.net2.bias", (C,)))}
    gamma = 1.0 + 0.2 * torch.from_numpy(weights.fill_like(f"ffb.{tag}.norm.weight", (C,), kind="unit"))
    beta = 0.1 * torch.from_numpy(weights.fill_like(f"ffb.{tag}.norm.bias", (C,), kind="unit"))
    return sd, gamma, beta


def _oracle(sd, gamma, beta, x):
    return x + unet_ref.feed_forward(sd, "f", unet_ref.layer_norm(x, gamma, beta), unet_ref._Mode("fp32"))


def _run_fused(sd, gamma, beta, x, want_stats=True):
    from leftrefill_amd import ops, packing
    d = dev()
    H = sd["f.net.2.weight"].shape[1]
    wf, bf, _cs = packing.fold_layernorm(sd["f.net.0.proj.weight"], sd["f.net.0.proj.bias"], gamma, beta)
    perm = packing.geglu_perm(H)
    w1, b1 = wf[perm].contiguous().to(d), bf[perm].contiguous().to(d)
    w2 = packing.pack_pieces(sd["f.net.2.weight"]).to(d)
    return ops.ffn_block(x.reshape(-1, C).half().to(d), w1, b1, w2, sd["f.net.2.bias"].to(d), eps=1e-5, want_stats=want_stats)


@pytest.mark.parametrize("M,H", [(128, 1280), (384, 1280), (256, 64), (128, 2048), (256, 192)])
def test_ffn_block_vs_oracle(M, H):
    sd, gamma, beta = _params(f"p{H}", H)
    x = h16(G.T(f"ffb.{M}.{H}.x", (M, C)) * 1.3 + 0.2)
    ref = _oracle(sd, gamma, beta, x)
    out, st = _run_fused(sd, gamma, beta, x)
    # two chained products with an fp16 hand-off of the gated hidden activation, LayerNorm folded into fp16 weights
    report(f"ffn M{M} H{H}", out, ref, rtol=3e-3, atol=3e-3)
    o32 = out.float()
    assert st.shape == (M, 2, 2)          # (sum, sumsq) partials of the two column halves
    torch.testing.assert_close(st[:, :, 0].sum(1), o32.sum(1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(st[:, :, 1].sum(1), (o32 * o32).sum(1), rtol=1e-5, atol=1e-3)


def test_ffn_block_hot_shape_and_reruns():
    """configs[1] shape of the level-0 blocks: M = 8 x 8192 rows, H = 1280 (sampled rows against the oracle), bit-identical reruns."""
    M, H = 65536, 1280
    sd, gamma, beta = _params("hot", H)
    g = torch.Generator().manual_seed(11)
    x = h16(torch.randn(M, C, generator=g))
    out = _run_fused(sd, gamma, beta, x, want_stats=False)
    out2 = _run_fused(sd, gamma, beta, x, want_stats=False)
    assert torch.equal(out, out2)
    rows = torch.arange(0, M, 61)
    report("ffn hot", out[rows], _oracle(sd, gamma, beta, x[rows]), rtol=3e-3, atol=3e-3)


def test_ffn_unsupported_shapes_are_reported():
    from leftrefill_amd import ops
    assert not ops.ffn_ok(100, 320, 1280) and not ops.ffn_ok(256, 640, 2560) and not ops.ffn_ok(256, 320, 2112)
    assert ops.ffn_ok(65536, 320, 1280)


def test_engine_transformer_block_fused_ffn_equals_unfused():
    """Drop-in BasicTransformerBlock at C = 320 through the engine with the fused feed-forward block on / off."""
    import importlib
    from leftrefill_amd import engine
    from leftrefill_amd.dropin import install
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(0)
    d = dev()
    blk = att.BasicTransformerBlock(320, 5, 64, context_dim=1024).to(d).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn_like(p_) * 0.05)
        for n_ in (blk.norm1, blk.norm2, blk.norm3):
            n_.weight.add_(1.0)
    x = torch.randn(2, 256, 320, device=d)
    ctx = torch.randn(2, 77, 1024, device=d)
    outs = []
    for flag in (True, False):
        engine.FFN_FUSED = flag
        try:
            with torch.no_grad():
                outs.append(blk(x, context=ctx).float().cpu())
        finally:
            engine.FFN_FUSED = True
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"[fused vs unfused feed-forward in BasicTransformerBlock] max abs diff {err:.3e} at |out| {outs[1].abs().max().item():.2f}")
    assert err <= 1e-2 * max(1.0, outs[1].abs().max().item())


@pytest.mark.parametrize("M", [128, 512])
def test_ffn_block_with_fused_proj_out(M):
    """POST variant: out = (x + ff(LN x)) Wp^T + bp + x_in (SpatialTransformer.proj_out + residual, attention.py:412-419) in the same
    launch, plus the per-channel GroupNorm statistics of the output over blocks of 128 rows."""
    from leftrefill_amd import ops, packing
    d = dev()
    H = 1280
    sd, gamma, beta = _params("post", H)
    wp = h16(torch.from_numpy(weights.fill_like("ffb.post.proj_out.weight", (C, C))))
    bp = torch.from_numpy(weights.fill_like("ffb.post.proj_out.bias", (C,)))
    x = h16(G.T(f"ffb.post.{M}.x", (M, C)) * 1.3 + 0.2)
    x_in = h16(G.T(f"ffb.post.{M}.xin", (M, C)))
    x3 = _oracle(sd, gamma, beta, x)
    ref = x_in + torch.nn.functional.linear(x3, wp, bp)
    wf, bf, _cs = packing.fold_layernorm(sd["f.net.0.proj.weight"], sd["f.net.0.proj.bias"], gamma, beta)
    perm = packing.geglu_perm(H)
    out, (gs, rows, gp, chunks) = ops.ffn_block(x.half().to(d), wf[perm].contiguous().to(d), bf[perm].contiguous().to(d),
                                                packing.pack_pieces(sd["f.net.2.weight"]).to(d), sd["f.net.2.bias"].to(d), eps=1e-5,
                                                post=(packing.pack_pieces(wp).to(d), bp.to(d), x_in.half().to(d)), want_gn_stats=True,
                                                gn_hw=128 if M == 128 else 256)
    report(f"ffn+proj_out M{M}", out, ref, rtol=3e-3, atol=5e-3)
    assert rows == 128 and gs.shape == (M // 128, C, 2)
    # per-group sums (10 channels per group) of every 128-row block, laid out [sample][chunk][32][2] for lr_groupnorm_apply_n
    hw = 128 if M == 128 else 256
    assert chunks == hw // 128 and gp.shape == (M // hw, chunks, 32, 2)
    og = out.double().reshape(M // hw, chunks, 128, 32, C // 32)
    torch.testing.assert_close(gp[..., 0].double(), og.sum((2, 4)), rtol=1e-5, atol=5e-3)
    torch.testing.assert_close(gp[..., 1].double(), (og * og).sum((2, 4)), rtol=1e-5, atol=5e-3)
    o32 = out.float().reshape(M // 128, 128, C)
    torch.testing.assert_close(gs[:, :, 0], o32.sum(1), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(gs[:, :, 1], (o32 * o32).sum(1), rtol=1e-5, atol=2e-3)
    out2 = ops.ffn_block(x.half().to(d), wf[perm].contiguous().to(d), bf[perm].contiguous().to(d),
                         packing.pack_pieces(sd["f.net.2.weight"]).to(d), sd["f.net.2.bias"].to(d), eps=1e-5,
                         post=(packing.pack_pieces(wp).to(d), bp.to(d), x_in.half().to(d)))
    assert torch.equal(out, out2)


def test_spatial_transformer_post_fused_equals_separate_proj_out():
    """Drop-in SpatialTransformer at C = 320 through engine.spatial_transformer (with the per-context K / V operands) with proj_out
    fused behind the feed-forward vs the separate GEMM."""
    import importlib
    from leftrefill_amd import engine, ops
    from leftrefill_amd.dropin import install
    install()
    att = importlib.import_module("ldm.modules.attention")
    torch.manual_seed(2)
    d = dev()
    st = att.SpatialTransformer(320, 5, 64, depth=1, context_dim=1024, use_linear=True).to(d).eval()
    with torch.no_grad():
        for p_ in st.parameters():
            p_.copy_(torch.randn_like(p_) * 0.05)
        st.norm.weight.add_(1.0)
        for blk in st.transformer_blocks:
            for n_ in (blk.norm1, blk.norm2, blk.norm3):
                n_.weight.add_(1.0)
    ps = engine.PackedST(st)
    ps.blocks[0].kv_slot = 0
    N, H_, W_, Lc = 2, 16, 16, 77
    tok = torch.randn(N * H_ * W_, 320, device=d).half()
    ctx = torch.randn(N * Lc, 1024, device=d).half()
    pa = ps.blocks[0].attn2
    kv = ops.gemm_conv(ctx, pa.kv.w, B=1, H=1, W=N * Lc, taps=1)
    cache = [(kv, None, ops.gemm_conv(ctx, pa.xk, B=1, H=1, W=N * Lc, taps=1), ops.xattn_pack_vt(kv[:, 320:], N, 5, Lc))]
    outs, gss = [], []
    for flag in (True, False):
        engine.FFN_POST = flag
        try:
            with torch.no_grad():
                a_ = engine.spatial_transformer(engine.Act(tok, N, H_, W_), ctx, Lc, ps, cache)
            outs.append(a_.tok.float().cpu())
            gss.append(a_.gs)
        finally:
            engine.FFN_POST = True
    err = (outs[0] - outs[1]).abs().max().item()
    print(f"[proj_out fused behind the feed-forward vs separate GEMM] max abs diff {err:.3e} at |out| {outs[1].abs().max().item():.2f}")
    assert err <= 1e-2 * max(1.0, outs[1].abs().max().item())
    assert gss[0] is not None and gss[0][1] == 128      # GroupNorm statistics ride along (128-row blocks)
