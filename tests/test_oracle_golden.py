"""CPU: the oracle restatement (oracle/) against golden vectors generated from the real reference.

Tolerances are fp32-vs-fp32 with a different summation order: 1e-5 abs per operator,
1e-4 on whole-network outputs.
"""
import numpy as np
import pytest
import torch

from oracle import ddim_ref, golden_spec as G, unet_ref, weights


@pytest.mark.parametrize("name,kind,p", G.OP_CASES, ids=[c[0] for c in G.OP_CASES])
def test_op_matches_reference(golden, name, kind, p):
    ref = golden("ops")[name]
    out = G.op_oracle(name, kind, p).numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-5)


def test_timestep_embedding(golden):
    t = torch.tensor([1, 21, 481, 981])
    for dim in (320, 64):
        ref = golden("ops")[f"timestep_embedding_{dim}"]
        out = unet_ref.timestep_embedding(t, dim).numpy()
        # cos/sin of fp32 arguments up to 981 rad: identical formula, allow 1 ulp of libm difference
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6)
        assert np.allclose(out[:, 0], np.cos(t.numpy().astype(np.float32)), atol=1e-6)  # cos first


@pytest.mark.parametrize("S", [10, 50])
@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_schedule_tables_bit_exact(golden, S, eta):
    g = golden("sampler")
    tabs = ddim_ref.ddim_tables(S, eta)
    tag = f"sched_S{S}_eta{int(eta)}"
    assert np.array_equal(tabs["timesteps"], g[tag + ".timesteps"])
    for k in ("alphas", "alphas_prev", "sigmas", "sqrt_one_minus_alphas"):
        assert np.array_equal(tabs[k], g[f"{tag}.{k}"]), k
    assert np.array_equal(ddim_ref.alphas_cumprod(), g["alphas_cumprod"])
    if S == 50:
        assert tabs["timesteps"][0] == 1 and tabs["timesteps"][1] == 21 and tabs["timesteps"][-1] == 981
    else:
        assert list(tabs["timesteps"][:2]) == [1, 101] and tabs["timesteps"][-1] == 901


@pytest.mark.parametrize("case,S,eta,index", G.STEP_CASES, ids=[c[0] for c in G.STEP_CASES])
def test_single_ddim_step(golden, case, S, eta, index):
    g = golden("sampler")
    B, h, w = 2, 8, 16
    x = G.T(case + ".x", (B, 4, h, w))
    e = G.T(case + ".e", (2 * B, 4, h, w))
    noise = G.T(case + ".noise", (B, 4, h, w))
    tabs = ddim_ref.ddim_tables(S, eta)
    e_u, e_c = e.chunk(2)
    x_prev, pred_x0 = ddim_ref.cfg_ddim_update(x, e_u, e_c, G.CFG_SCALE, tabs["alphas"][index],
                                               tabs["alphas_prev"][index], tabs["sigmas"][index],
                                               tabs["sqrt_one_minus_alphas"][index], noise)
    np.testing.assert_allclose(x_prev.numpy(), g[case + ".x_prev"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pred_x0.numpy(), g[case + ".pred_x0"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case,S,eta,B,h,w", G.TRAJ_CASES, ids=[c[0] for c in G.TRAJ_CASES])
def test_ddim_trajectory(golden, case, S, eta, B, h, w):
    g = golden("sampler")
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    sd = G.unet_state(G.TRAJ_CONFIG)
    x_T = G.T(case + ".x_T", (B, 4, h, w))
    c_concat = G.T(case + ".c_concat", (B, 5, h, w))
    c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim))
    uc_cross = G.T(case + ".uc_cross", (B, 77, cfg.context_dim))
    noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)) for i in range(S)]
    t_seq = []

    def apply(xc, t, ctx):
        t_seq.append(int(t[0]))
        assert torch.all(t == t[0]) and xc.shape[0] == 2 * B
        return unet_ref.unet_forward(sd, cfg, xc, t, ctx)

    trace = []
    samples, inter = ddim_ref.ddim_sample(apply, S, x_T, c_concat, c_cross, uc_cross, G.CFG_SCALE, eta=eta,
                                          noises=noises, trace=trace)
    # bit-identical step indexing
    assert t_seq == list(g[case + ".t_seq"])
    assert [tr[2] for tr in trace] == list(range(S - 1, -1, -1))
    ref = g[case + ".samples"]
    # random-weight UNet makes the trajectory expansive (|x| grows to ~10-40): compare relative to its scale
    scale = np.abs(ref).max()
    assert np.abs(samples.numpy() - ref).max() <= 2e-4 * scale
    assert len(inter["x_inter"]) == g[case + ".x_inter"].shape[0]
    np.testing.assert_allclose(torch.stack(inter["pred_x0"]).numpy(), g[case + ".pred_x0"], rtol=0, atol=2e-4 * scale)


@pytest.mark.parametrize("case,S,eta,B,h,w,K,seed", G.MULTI_CASES, ids=[c[0] for c in G.MULTI_CASES])
def test_ddim_multi_condition_trajectory(golden, case, S, eta, B, h, w, K, seed):
    """ddim_multi_sampling (reference ddim.py:147-222) incl. the python-`random` choice of the shared right half."""
    import random
    g = golden("sampler_multi")
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    sd = G.unet_state(G.TRAJ_CONFIG)
    x_T = [G.T(f"{case}.x_T{k}", (B, 4, h, w)) for k in range(K)]
    cc = [G.T(f"{case}.c_concat{k}", (B, 5, h, w)) for k in range(K)]
    c = [G.T(f"{case}.c_cross{k}", (B, 77, cfg.context_dim)) for k in range(K)]
    uc = [G.T(f"{case}.uc_cross{k}", (B, 77, cfg.context_dim)) for k in range(K)]
    noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)) for i in range(S * K)]
    t_seq = []

    def apply(xc, t, ctx):
        t_seq.append(int(t[0]))
        return unet_ref.unet_forward(sd, cfg, xc, t, ctx)

    random.seed(seed)
    out = ddim_ref.ddim_multi_sample(apply, S, x_T, cc, c, uc, G.CFG_SCALE, eta=eta, noises=noises)
    assert t_seq == list(g[case + ".t_seq"])
    ref = g[case + ".samples"]
    assert np.abs(out.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("case,B,h,w,ts", G.TRAIN_CASES, ids=[c[0] for c in G.TRAIN_CASES])
def test_training_loss_and_context_gradient(golden, case, B, h, w, ts):
    """p_losses + backward of the real reference (through its CheckpointFunction) vs autograd on the restatement: pins the
    checker that tests/test_gpu_backward.py holds the HIP backward kernels against."""
    g = golden("train")
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    sd = G.unet_state(G.TRAJ_CONFIG)
    x_start = G.T(case + ".x_start", (B, 4, h, w))
    noise = G.T(case + ".noise", (B, 4, h, w))
    c_concat = G.T(case + ".c_concat", (B, 5, h, w))
    c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim)).requires_grad_(True)
    t = torch.tensor(ts, dtype=torch.long)
    with torch.enable_grad():
        loss, loss_simple, x_noisy = ddim_ref.p_losses(
            lambda xc, tt, ctx: unet_ref.unet_forward.__wrapped__(sd, cfg, xc, tt, ctx), x_start, c_concat, c_cross, t, noise)
        loss.backward()
    np.testing.assert_allclose(x_noisy.numpy(), g[case + ".x_noisy"], atol=1e-6)
    np.testing.assert_allclose(loss.item(), g[case + ".loss"], rtol=1e-5)
    np.testing.assert_allclose(loss_simple.mean().item(), g[case + ".loss_simple"], rtol=1e-5)
    ref = g[case + ".dctx"]
    assert np.abs(c_cross.grad.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize("case,V,concat,b,H,W", G.MV_CASES, ids=[c[0] for c in G.MV_CASES])
def test_multiview_unet(golden, case, V, concat, b, H, W):
    cfg = G.mv_config(V, concat)
    sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
    n = b * (V - 1 if concat else V)
    x, t, ctx = G.unet_inputs(case, cfg, n, H, W, [501] * n)
    out = unet_ref.unet_forward(sd, cfg, x, t, ctx).numpy()
    np.testing.assert_allclose(out, golden("multiview")[case], rtol=1e-4, atol=1e-4)


def _check_unet(golden, case, cname, N, H, W, ts, sd):
    cfg = G.CONFIGS[cname]
    x, t, ctx = G.unet_inputs(case, cfg, N, H, W, ts)
    taps = {}
    out = unet_ref.unet_forward(sd, cfg, x, t, ctx, taps=taps).numpy()
    g = golden("unet")
    np.testing.assert_allclose(out, g[case], rtol=1e-4, atol=1e-4)
    keys = list(g[case + ".tap_keys"])
    stats = g[case + ".tap_stats"]
    for k, (mean, std, amax) in zip(keys, stats):
        v = taps[str(k)]
        assert abs(v.mean().item() - mean) < 1e-4 + 1e-4 * abs(mean), k
        assert abs(v.std().item() - std) < 1e-4 * std + 1e-5, k


@pytest.mark.parametrize("which", ["SMALL", "MID"])
def test_unet_reduced_width(golden, which):
    sd = G.unet_state(which)
    for case, cname, N, H, W, ts in G.UNET_CASES:
        if cname == which:
            _check_unet(golden, case, cname, N, H, W, ts, sd)


def test_unet_full_width(golden):
    """866 M-parameter SD2-inpainting UNet (shipped config) at latent 8x16 and 16x32; ~1.5 min (weight fill)."""
    sd = G.unet_state("FULL")
    assert len(sd) == 686 and sum(v.numel() for v in sd.values()) == 865_925_124
    for case, cname, N, H, W, ts in G.UNET_CASES:
        if cname == "FULL":
            _check_unet(golden, case, cname, N, H, W, ts, sd)


def test_flop_model_matches_survey():
    assert abs(unet_ref.flops_per_sample(unet_ref.FULL, 64, 128) / 1e9 - 1849.75) < 0.01
    assert abs(unet_ref.flops_per_sample(unet_ref.FULL, 32, 64) / 1e9 - 373.56) < 0.01


def test_autocast16_emulation_is_close_to_fp32():
    """The fp16-rounding emulation stays within fp16 noise of the fp32 oracle (sizes the HIP tolerance)."""
    cfg = unet_ref.SMALL
    sd = G.unet_state("SMALL")
    x, t, ctx = G.unet_inputs("unet_small_16x32_b4", cfg, 4, 16, 32, [481, 481, 21, 21])
    a = unet_ref.unet_forward(sd, cfg, x, t, ctx)
    b = unet_ref.unet_forward(sd, cfg, x, t, ctx, mode="autocast16")
    rel = ((a - b).norm() / a.norm()).item()
    assert 1e-5 < rel < 1e-2


def test_oracle_attention_fused_form_matches_materialised_form():
    """oracle.unet_ref.ATTENTION_IMPL = "sdpa" (the reference's xformers path, attention.py:199-250) is the same function as the
    materialised-logits form (165-196) that pins the parity tests: bench.py's cpu_baseline may time either."""
    import torch
    from oracle import unet_ref
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(2, n_, 5 * 64, generator=g) for n_ in (300, 77, 77))
    m = unet_ref._Mode("fp32")
    ref = unet_ref.attention(q, k, v, 5, m)
    prev = unet_ref.ATTENTION_IMPL
    try:
        unet_ref.ATTENTION_IMPL = "sdpa"
        out = unet_ref.attention(q, k, v, 5, m)
        emu = unet_ref.attention(q, k, v, 5, unet_ref._Mode("autocast16"))      # the emulation keeps the materialised form
    finally:
        unet_ref.ATTENTION_IMPL = prev
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(emu, unet_ref.attention(q, k, v, 5, unet_ref._Mode("autocast16")))
