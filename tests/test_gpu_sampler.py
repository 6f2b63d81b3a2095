"""GPU parity of the sampler path: DDIMSampler + LatentInpaintDiffusion.apply_model + hybrid wrapper + UNet (MID width)
against the reference trajectories (golden G7) -- pins the step indexing bit-exactly and the latents within fp16 noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ddim_ref, golden_spec as G, unet_ref  # noqa: E402


def build(dev):
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    m = RefInpaintLDM(first_stage_config={"target": "torch.nn.Identity"},
                      cond_stage_config={"target": "torch.nn.Identity"},
                      unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                   "params": cfg.kwargs()},
                      conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                      timesteps=1000, channels=4, data_config={"img_size": 256})
    m.model.diffusion_model.load_state_dict(G.unet_state(G.TRAJ_CONFIG), strict=True)
    return m.to(dev).eval(), cfg


_cache = {}


def model(dev):
    if "m" not in _cache:
        _cache["m"] = build(dev)
    return _cache["m"]


@pytest.mark.parametrize("case,S,eta,B,h,w", G.TRAJ_CASES, ids=[c[0] for c in G.TRAJ_CASES])
def test_trajectory(golden, case, S, eta, B, h, w):
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    g = golden("sampler")
    x_T = G.T(case + ".x_T", (B, 4, h, w)).to(dev)
    c_concat = G.T(case + ".c_concat", (B, 5, h, w)).to(dev)
    c_cross = G.T(case + ".c_cross", (B, 77, cfg.context_dim)).to(dev)
    uc_cross = G.T(case + ".uc_cross", (B, 77, cfg.context_dim)).to(dev)
    noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)).to(dev) for i in range(S)]
    import ldm.models.diffusion.ddim as ddim_mod
    it = iter(noises)
    orig_noise = ddim_mod.noise_like
    ddim_mod.noise_like = lambda shape, device, repeat=False: next(it)
    t_seq = []
    orig_apply = m.apply_model

    def spy(x, t, c, **kw):
        t_seq.append(int(t[0].item()))
        assert x.shape[0] == 2 * B and torch.all(t == t[0])
        return orig_apply(x, t, c, **kw)

    m.apply_model = spy
    try:
        cond = {"c_concat": [c_concat], "c_crossattn": [c_cross]}
        uc = {"c_concat": [c_concat], "c_crossattn": [uc_cross]}
        samples, inter = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=S, eta=eta, x_T=x_T,
                                      unconditional_guidance_scale=G.CFG_SCALE, unconditional_conditioning=uc)
    finally:
        ddim_mod.noise_like = orig_noise
        m.apply_model = orig_apply
    assert t_seq == list(g[case + ".t_seq"]), "DDIM step indexing must be bit-identical"
    ref = torch.from_numpy(g[case + ".samples"])
    assert len(inter["x_inter"]) == g[case + ".x_inter"].shape[0]
    scale = ref.abs().max().item()
    err = (samples.float().cpu() - ref).abs()
    # the random-weight UNet makes the S-step map expansive (|x| grows to 10-40), so compare relative to the
    # trajectory's scale; the budget is S accumulated fp16 UNet evaluations (~2e-3 relative each, see test_gpu_unet)
    # measured against what the oracle's own fp16 emulation drifts by on the same trajectory.
    sd = G.unet_state(G.TRAJ_CONFIG)
    emul, _ = ddim_ref.ddim_sample(lambda xc, t, ctx: unet_ref.unet_forward(sd, cfg, xc, t, ctx, mode="autocast16"),
                                   S, x_T.cpu(), c_concat.cpu(), c_cross.cpu(), uc_cross.cpu(), G.CFG_SCALE, eta=eta,
                                   noises=[n.cpu() for n in noises])
    err_e = (emul - ref).abs()
    rel, rel_e = (err.norm() / ref.norm()).item(), (err_e.norm() / ref.norm()).item()
    print(f"[traj {case}] max_abs {err.max().item():.3e} rel_l2 {rel:.3e} | autocast16 emulation max_abs "
          f"{err_e.max().item():.3e} rel_l2 {rel_e:.3e} | scale {scale:.2f}")
    assert torch.isfinite(samples).all()
    assert rel <= max(2.0 * rel_e, 5e-3), (rel, rel_e)


def test_sample_is_deterministic_and_graph_reused():
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    B, h, w = 2, 8, 16
    x_T = G.T("det.x_T", (B, 4, h, w)).to(dev)
    cond = {"c_concat": [G.T("det.cc", (B, 5, h, w)).to(dev)], "c_crossattn": [G.T("det.c", (B, 77, cfg.context_dim)).to(dev)]}
    uc = {"c_concat": cond["c_concat"], "c_crossattn": [G.T("det.uc", (B, 77, cfg.context_dim)).to(dev)]}
    outs = []
    for _ in range(2):
        s, _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=5, eta=0.0, x_T=x_T,
                            unconditional_guidance_scale=2.5, unconditional_conditioning=uc)
        outs.append(s)
    assert torch.equal(outs[0], outs[1])
    assert len(m.model.diffusion_model._graphs) >= 1


def test_sampler_timestep_table_is_bit_identical(monkeypatch):
    """DDIMSampler has the UNet compute the embedding rows (time MLP + emb_layers, reference openaimodel.py:775-776, 266) of all
    its timesteps before the loop and names each step's timestep on the host: same samples, bit for bit, as computing them from
    the `timesteps` tensor in every step -- in the captured-graph mode and in eager mode -- and the table is consulted at all."""
    from ldm.modules.diffusionmodules import openaimodel as om
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    B, h, w = 1, 8, 16
    x_T = G.T("tt.x_T", (B, 4, h, w)).to(dev)
    cond = {"c_concat": [G.T("tt.cc", (B, 5, h, w)).to(dev)], "c_crossattn": [G.T("tt.c", (B, 77, cfg.context_dim)).to(dev)]}
    uc = {"c_concat": cond["c_concat"], "c_crossattn": [G.T("tt.uc", (B, 77, cfg.context_dim)).to(dev)]}
    unet = m.model.diffusion_model
    outs = {}
    try:
        for graph in (True, False):
            unet.use_hip_graph = graph
            for table in (True, False):
                monkeypatch.setattr(om, "EMB_TABLE", table)
                unet._graphs.clear()
                torch.manual_seed(7)
                outs[graph, table], _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=20, eta=1.0, x_T=x_T,
                                                    unconditional_guidance_scale=2.5, unconditional_conditioning=uc)
                assert (len(unet._emb_table) == 20) == table
                assert unet._t_host is None
        ref = outs[True, False]
        assert all(torch.equal(o, ref) for o in outs.values())
        # a caller that does not go through the sampler (no host hint) gets the embedding of ITS timesteps, table or not
        monkeypatch.setattr(om, "EMB_TABLE", True)
        unet.prepare_timesteps([999, 500])
        x = torch.cat([x_T, x_T])
        ctx = torch.cat([uc["c_crossattn"][0], cond["c_crossattn"][0]]).half()
        xin = torch.cat([x, torch.cat([cond["c_concat"][0]] * 2)], 1)
        e1 = unet(xin, torch.full((2,), 321, device=dev), ctx)
        unet._emb_table = {}
        e2 = unet(xin, torch.full((2,), 321, device=dev), ctx)
        assert torch.equal(e1, e2)
        # a UNet call made DURING a sampling, outside p_sample_ddim (a callback, a corrector, a second sampler sharing the
        # model), with a timestep of its own must not pick up the step's precomputed row: the hint is scoped to p_sample_ddim's calls
        seen = []

        def cb(pred_x0, i):
            assert unet._t_host is None
            seen.append(unet(xin, torch.full((2,), 321, device=dev), ctx))

        torch.manual_seed(7)
        out_cb, _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=20, eta=1.0, x_T=x_T, unconditional_guidance_scale=2.5,
                                 unconditional_conditioning=uc, img_callback=cb)
        assert len(seen) == 20 and all(torch.equal(e_, e1) for e_ in seen)
        assert torch.equal(out_cb, ref)
        # a re-pack (weights changed) drops the rows of the old packing
        unet.prepare_timesteps([999, 500])
        unet.prepare(force=True)
        unet._t_host = 999
        try:
            unet._emb_rows(torch.full((2,), 999, device=dev), 2)
            assert unet._emb_table == {}
        finally:
            unet._t_host = None
    finally:
        unet.use_hip_graph = True
        unet._emb_table = {}
        unet._graphs.clear()


def test_sampler_shared_prefix_matches_full_cfg(monkeypatch):
    """DDIMSampler sets UNetModel.cfg_shared_prefix for its [x; x] CFG batches when the non-context conditioning of uncond
    and cond is identical: same samples, bit for bit, as with LEFTREFILL_CFG_SHARED_PREFIX=0; and it does NOT set it when
    c_concat differs between the two."""
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    B, h, w = 2, 8, 16
    x_T = G.T("shp.x_T", (B, 4, h, w)).to(dev)
    cond = {"c_concat": [G.T("shp.cc", (B, 5, h, w)).to(dev)], "c_crossattn": [G.T("shp.c", (B, 77, cfg.context_dim)).to(dev)]}
    uc = {"c_concat": [cond["c_concat"][0].clone()], "c_crossattn": [G.T("shp.uc", (B, 77, cfg.context_dim)).to(dev)]}
    unet = m.model.diffusion_model
    seen = []
    orig = unet._run_plan

    def spy(x, t, c, kv=None, shared_prefix=False, **kw):
        seen.append(bool(shared_prefix))
        return orig(x, t, c, kv, shared_prefix, **kw)
    unet._run_plan = spy
    try:
        outs = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("LEFTREFILL_CFG_SHARED_PREFIX", flag)
            unet._graphs.clear()
            seen.clear()
            outs[flag], _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=5, eta=0.0, x_T=x_T,
                                         unconditional_guidance_scale=2.5, unconditional_conditioning=uc)
            assert seen and all(s_ == (flag == "1") for s_ in seen), (flag, seen)
        assert torch.equal(outs["1"], outs["0"])
        monkeypatch.setenv("LEFTREFILL_CFG_SHARED_PREFIX", "1")
        uc2 = {"c_concat": [cond["c_concat"][0] + 0.5], "c_crossattn": uc["c_crossattn"]}
        unet._graphs.clear()
        seen.clear()
        m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=4, eta=0.0, x_T=x_T, unconditional_guidance_scale=2.5,
                     unconditional_conditioning=uc2)
        assert seen and not any(seen)
    finally:
        unet._run_plan = orig
        unet._graphs.clear()


def test_sampler_split_cfg_matches_joint_cfg():
    """Split classifier-free guidance (leftrefill_amd.dist.enable_split_cfg: the unconditional and the conditional pass run as
    two UNet calls of batch B -- on two ranks when a process group exists, one after the other here) against the reference's
    joint [uncond; cond] batch of 2B (ddim.py:317-343): the same samples up to fp16 noise (different tile plans for M and 2M)."""
    from leftrefill_amd import dist as lrd
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    B, h, w = 2, 16, 32
    x_T = G.T("spl.x_T", (B, 4, h, w)).to(dev)
    cond = {"c_concat": [G.T("spl.cc", (B, 5, h, w)).to(dev)], "c_crossattn": [G.T("spl.c", (B, 77, cfg.context_dim)).to(dev)]}
    uc = {"c_concat": [cond["c_concat"][0].clone()], "c_crossattn": [G.T("spl.uc", (B, 77, cfg.context_dim)).to(dev)]}
    calls = []
    orig_apply = m.apply_model

    def spy(x, t, c, **kw):
        calls.append(x.shape[0])
        return orig_apply(x, t, c, **kw)
    m.apply_model = spy
    try:
        joint, _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=5, eta=0.0, x_T=x_T,
                                unconditional_guidance_scale=2.5, unconditional_conditioning=uc)
        assert calls == [2 * B] * 5
        calls.clear()
        lrd.enable_split_cfg(True)
        try:
            split, _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=5, eta=0.0, x_T=x_T,
                                    unconditional_guidance_scale=2.5, unconditional_conditioning=uc)
        finally:
            lrd.enable_split_cfg(False)
        assert calls == [B] * 10
    finally:
        m.apply_model = orig_apply
    rel = ((split - joint).norm() / joint.norm()).item()
    print(f"[split cfg vs joint cfg] rel_l2 {rel:.3e}")
    assert torch.isfinite(split).all() and rel < 2e-2


def test_log_images_end_to_end_glue():
    """RefInpaintLDM.log_images (ref_inpainting_ldm.py:37-72): VAE encode of image / masked image, nearest mask
    down-sampling, channel order [z | mask | masked latent], unconditional prompt, 50->5 step CFG sampling, VAE decode.
    log_images runs the VAE on the HIP kernels; the expected value uses the PyTorch definition of the same module
    (pinned to the reference on CPU) and the CPU oracle for the sampler / UNet."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM
    from oracle import weights
    dev = torch.device("cuda:0")
    cfg = G.CONFIGS[G.TRAJ_CONFIG]
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    m = RefInpaintLDM(first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                          "params": {"ddconfig": dd, "embed_dim": 4,
                                                     "lossconfig": {"target": "torch.nn.Identity"}}},
                      cond_stage_config={"target": "torch.nn.Identity"},
                      unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                   "params": cfg.kwargs()},
                      conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                      timesteps=1000, channels=4, first_stage_key="image", cond_stage_key="txt",
                      cond_stage_trainable=True, data_config={"img_size": 64})
    m.model.diffusion_model.load_state_dict(G.unet_state(G.TRAJ_CONFIG), strict=True)
    m.first_stage_model.load_state_dict({k: torch.from_numpy(weights.fill_like("vae2." + k, v.shape))
                                         for k, v in m.first_stage_model.state_dict().items()})
    B, S = 2, 64
    ctx_c = G.T("e2e.c", (B, 77, cfg.context_dim))
    ctx_u = G.T("e2e.u", (B, 77, cfg.context_dim))

    class Prompt(torch.nn.Module):
        def encode(self, txt):
            assert isinstance(txt, list) and len(txt) == B
            return (ctx_u if txt[0] == "" else ctx_c).to(dev)

    m.cond_stage_model = Prompt()
    m = m.to(dev).eval()
    img = G.T("e2e.img", (B, S, 2 * S, 3), "unit")
    mask = torch.zeros(B, S, 2 * S, 1)
    mask[:, 16:48, S + 8:S + 40] = 1.0
    batch = {"image": img.to(dev), "mask": mask.to(dev), "masked_image": (img * (mask < 0.5)).to(dev), "txt": ["p"] * B}
    x_T = G.T("e2e.x_T", (B, 4, S // 8, 2 * S // 8))
    import ldm.models.diffusion.ddim as ddim_mod
    orig_sampling = ddim_mod.DDIMSampler.ddim_sampling

    def with_xT(self, cond, shape, **kw):
        kw["x_T"] = x_T.to(dev)
        return orig_sampling(self, cond, shape, **kw)

    ddim_mod.DDIMSampler.ddim_sampling = with_xT
    try:
        out = m.log_images(batch, B, ddim_steps=5, ddim_eta=0.0, unconditional_guidance_scale=2.5)
    finally:
        ddim_mod.DDIMSampler.ddim_sampling = orig_sampling
    assert set(out) == {"masked_image", "origin_image", "pred"}
    assert out["pred"].shape == (B, 3, S, 2 * S) and torch.isfinite(out["pred"]).all()
    assert torch.equal(out["origin_image"], batch["image"].permute(0, 3, 1, 2))
    # expected value: same VAE (host torch) + CPU oracle for the sampler/UNet
    vae = m.first_stage_model
    vae.use_hip = False
    with torch.no_grad():
        lat = vae.encode(batch["masked_image"].permute(0, 3, 1, 2).float()).sample() * 0.18215
        mk = torch.nn.functional.interpolate(batch["mask"].permute(0, 3, 1, 2).float(), size=lat.shape[-2:])
        c_concat = torch.cat([mk, lat], 1).cpu()
        sd = G.unet_state(G.TRAJ_CONFIG)
        z, _ = ddim_ref.ddim_sample(lambda xc, t, c: unet_ref.unet_forward(sd, cfg, xc, t, c), 5, x_T, c_concat, ctx_c,
                                    ctx_u, 2.5, eta=0.0)
        ref = vae.decode((z / 0.18215).to(dev)).cpu()
    err = (out["pred"].float().cpu() - ref).abs().max().item()
    print(f"[log_images] max|pred - expected| = {err:.3e} (ref absmax {ref.abs().max().item():.2f})")
    assert err <= 5e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case,S,eta,B,h,w,K,seed", G.MULTI_CASES, ids=[c[0] for c in G.MULTI_CASES])
def test_multi_condition_sampler(golden, case, S, eta, B, h, w, K, seed):
    """DDIMSampler.sample with a LIST of conditionings -> ddim_multi_sampling (reference ddim.py:147-222), golden from
    the real reference incl. python-`random` pick sequence; timestep sequence bit-identical."""
    import random
    dev = torch.device("cuda:0")
    m, cfg = model(dev)
    g = golden("sampler_multi")
    x_T = [G.T(f"{case}.x_T{k}", (B, 4, h, w)).to(dev) for k in range(K)]
    conds = [{"c_concat": [G.T(f"{case}.c_concat{k}", (B, 5, h, w)).to(dev)],
              "c_crossattn": [G.T(f"{case}.c_cross{k}", (B, 77, cfg.context_dim)).to(dev)]} for k in range(K)]
    ucs = [{"c_concat": conds[k]["c_concat"], "c_crossattn": [G.T(f"{case}.uc_cross{k}", (B, 77, cfg.context_dim)).to(dev)]}
           for k in range(K)]
    noises = [G.T(f"{case}.noise{i}", (B, 4, h, w)) for i in range(S * K)]
    import ldm.models.diffusion.ddim as ddim_mod
    from ldm.models.diffusion.ddim import DDIMSampler
    it = iter(noises)
    orig_noise = ddim_mod.noise_like
    ddim_mod.noise_like = lambda shape, device, repeat=False: next(it).to(device)
    t_seq = []
    orig_apply = m.apply_model

    def spy(x, t, c, **kw):
        t_seq.append(int(t[0].item()))
        return orig_apply(x, t, c, **kw)

    m.apply_model = spy
    try:
        random.seed(seed)
        samples, _ = DDIMSampler(m).sample(S, B, (4, h, w), conds, verbose=False, eta=eta, x_T=x_T,
                                           unconditional_guidance_scale=G.CFG_SCALE, unconditional_conditioning=ucs)
    finally:
        ddim_mod.noise_like = orig_noise
        m.apply_model = orig_apply
    assert t_seq == list(g[case + ".t_seq"])
    ref = torch.from_numpy(g[case + ".samples"])
    sd = G.unet_state(G.TRAJ_CONFIG)
    random.seed(seed)
    emul = ddim_ref.ddim_multi_sample(lambda xc, t, ctx: unet_ref.unet_forward(sd, cfg, xc, t, ctx, mode="autocast16"),
                                      S, [x.cpu() for x in x_T], [c["c_concat"][0].cpu() for c in conds],
                                      [c["c_crossattn"][0].cpu() for c in conds], [u["c_crossattn"][0].cpu() for u in ucs],
                                      G.CFG_SCALE, eta=eta, noises=noises)
    rel = ((samples.float().cpu() - ref).norm() / ref.norm()).item()
    rel_e = ((emul - ref).norm() / ref.norm()).item()
    print(f"[multi {case}] rel_l2 {rel:.3e} | autocast16 emulation {rel_e:.3e} | scale {ref.abs().max().item():.2f}")
    assert torch.isfinite(samples).all()
    assert rel <= max(2.0 * rel_e, 5e-3), (rel, rel_e)


@pytest.mark.parametrize("V,concat", [(3, True), (2, False)], ids=["v3_concat_target", "v2_plain"])
def test_multiview_log_images_end_to_end(V, concat):
    """`inpainting_ldm.multiview_ref_inpainting_ldm.RefInpaintLDM.log_images` (reference 113-180) over MultiViewUnetModel:
    5-D batch -> (b v) canvases -> joint sampling -> target view / reference slicing.  Expected value: PyTorch definition
    of the VAE + CPU oracle of the multi-view UNet and sampler."""
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.multiview_ref_inpainting_ldm import RefInpaintLDM
    from oracle import weights
    dev = torch.device("cuda:0")
    cfg = G.mv_config(V, concat)
    v = V - 1 if concat else V
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    m = RefInpaintLDM(first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                          "params": {"ddconfig": dd, "embed_dim": 4,
                                                     "lossconfig": {"target": "torch.nn.Identity"}}},
                      cond_stage_config={"target": "torch.nn.Identity"},
                      unet_config={"target": "ldm.modules.diffusionmodules.multiview_unet.MultiViewUnetModel",
                                   "params": cfg.kwargs()},
                      conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                      timesteps=1000, channels=4, first_stage_key="image", cond_stage_key="txt",
                      cond_stage_trainable=True, data_config={"img_size": 64}, view_mode=True, view_num=V,
                      concat_target=concat)
    assert (m.view_num, m.concat_target, m.view_mode) == (V, concat, True)
    sd = weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.")
    m.model.diffusion_model.load_state_dict(sd, strict=True)
    m.first_stage_model.load_state_dict({k: torch.from_numpy(weights.fill_like("vae2." + k, t.shape))
                                         for k, t in m.first_stage_model.state_dict().items()})
    b, S = 1, 64
    Wc = 2 * S if concat else S              # concat_target: [ref | target] canvases; otherwise square views
    n = b * v
    ctx_c = G.T("mve2e.c", (n, 77, cfg.context_dim))
    ctx_u = G.T("mve2e.u", (n, 77, cfg.context_dim))

    class Prompt(torch.nn.Module):
        def encode(self, txt):
            assert isinstance(txt, list) and len(txt) == n
            return (ctx_u if txt[0] == "" else ctx_c).to(dev)

    m.cond_stage_model = Prompt()
    m = m.to(dev).eval()
    img = G.T("mve2e.img", (b, v, S, Wc, 3), "unit")
    mask = torch.zeros(b, v, S, Wc, 1)
    mask[:, 0, 16:48, Wc - S + 8:Wc - S + 40] = 1.0
    batch = {"image": img.to(dev), "mask": mask.to(dev), "masked_image": (img * (mask < 0.5)).to(dev), "txt": ["p"] * n}
    x_T = G.T("mve2e.x_T", (n, 4, S // 8, Wc // 8))
    import ldm.models.diffusion.ddim as ddim_mod
    orig_sampling = ddim_mod.DDIMSampler.ddim_sampling

    def with_xT(self, cond, shape, **kw):
        kw["x_T"] = x_T.to(dev)
        return orig_sampling(self, cond, shape, **kw)

    ddim_mod.DDIMSampler.ddim_sampling = with_xT
    try:
        out = m.log_images(batch, ddim_steps=5, ddim_eta=0.0, unconditional_guidance_scale=2.5)
    finally:
        ddim_mod.DDIMSampler.ddim_sampling = orig_sampling
    assert set(out) == {"masked_image", "origin_image", "pred", "reference"}
    assert batch["image"].dim() == 4                       # flattened in place like the reference
    assert out["pred"].shape == (b, 3, S, S) and torch.isfinite(out["pred"]).all()
    flat_masked = batch["masked_image"].permute(0, 3, 1, 2).reshape(b, v, 3, S, Wc)
    if concat:
        assert torch.equal(out["reference"], flat_masked[..., :S]) and out["reference"].shape == (b, v, 3, S, S)
        assert torch.equal(out["origin_image"], batch["image"].permute(0, 3, 1, 2).reshape(b, v, 3, S, Wc)[:, 0, ..., S:])
    else:
        assert torch.equal(out["reference"], flat_masked[:, 1:])
    vae = m.first_stage_model
    vae.use_hip = False
    with torch.no_grad():
        lat = vae.encode(batch["masked_image"].permute(0, 3, 1, 2).float()).sample() * 0.18215
        mk = torch.nn.functional.interpolate(batch["mask"].permute(0, 3, 1, 2).float(), size=lat.shape[-2:])
        c_concat = torch.cat([mk, lat], 1).cpu()
        z, _ = ddim_ref.ddim_sample(lambda xc, t, c: unet_ref.unet_forward(sd, cfg, xc, t, c), 5, x_T, c_concat, ctx_c,
                                    ctx_u, 2.5, eta=0.0)
        full = vae.decode((z / 0.18215).to(dev)).cpu().reshape(b, v, 3, S, Wc)
    ref = full[:, 0, ..., S:] if concat else full[:, 0]
    err = (out["pred"].float().cpu() - ref).abs().max().item()
    print(f"[mv log_images V={V} concat={concat}] max|pred - expected| = {err:.3e} (ref absmax {ref.abs().max().item():.2f})")
    assert err <= 5e-2 * max(1.0, ref.abs().max().item())


def test_config1_full_width_256x512_10_steps():
    """BASELINE.json configs[0]: 1-ref inpainting at 256x512 (latent 32x64), bs=1, 10 DDIM steps, cfg=2.5 -- the full
    866 M-parameter SD2-inpainting UNet on the HIP path against the CPU oracle (fp32 restatement pinned to the reference
    at this width by goldens G4) on identical latents / timesteps.  Also prints the oracle's CPU time for this config."""
    import time
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM
    dev = torch.device("cuda:0")
    cfg = G.CONFIGS["FULL"]
    sd = G.unet_state("FULL")
    m = RefInpaintLDM(first_stage_config={"target": "torch.nn.Identity"},
                      cond_stage_config={"target": "torch.nn.Identity"},
                      unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                   "params": cfg.kwargs()},
                      conditioning_key="hybrid", scale_factor=0.18215, linear_start=0.00085, linear_end=0.0120,
                      timesteps=1000, channels=4, data_config={"img_size": 256})
    m.model.diffusion_model.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    B, h, w, S = 1, 32, 64, 10
    x_T = G.T("cfg1.x_T", (B, 4, h, w))
    c_concat = G.T("cfg1.c_concat", (B, 5, h, w))
    c_cross = G.T("cfg1.c_cross", (B, 77, cfg.context_dim))
    uc_cross = G.T("cfg1.uc_cross", (B, 77, cfg.context_dim))
    t_seq = []
    orig_apply = m.apply_model

    def spy(x, t, c, **kw):
        t_seq.append(int(t[0].item()))
        return orig_apply(x, t, c, **kw)

    m.apply_model = spy
    cond = {"c_concat": [c_concat.to(dev)], "c_crossattn": [c_cross.to(dev)]}
    uc = {"c_concat": [c_concat.to(dev)], "c_crossattn": [uc_cross.to(dev)]}
    samples, _ = m.sample_log(cond=cond, batch_size=B, ddim=True, ddim_steps=S, eta=0.0, x_T=x_T.to(dev),
                              unconditional_guidance_scale=G.CFG_SCALE, unconditional_conditioning=uc)
    m.apply_model = orig_apply
    trace = []
    t0 = time.time()
    ref, _ = ddim_ref.ddim_sample(lambda xc, t, ctx: unet_ref.unet_forward(sd, cfg, xc, t, ctx), S, x_T, c_concat,
                                  c_cross, uc_cross, G.CFG_SCALE, eta=0.0, trace=trace)
    cpu_s = time.time() - t0
    assert t_seq == [tr[1] for tr in trace] == list(range(901, 0, -100))       # [901, 801, ..., 1]
    err = (samples.float().cpu() - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"[config 1] 256x512 bs=1 S=10: rel_l2 {rel:.3e} max_abs {err.max().item():.3e} (|ref| max "
          f"{ref.abs().max().item():.2f}); CPU oracle {cpu_s:.1f} s on {torch.get_num_threads()} threads")
    assert torch.isfinite(samples).all()
    assert rel <= 6e-3
