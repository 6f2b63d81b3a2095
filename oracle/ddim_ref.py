"""CPU oracle for the DDIM / classifier-free-guidance sampler (test infrastructure).

Restates, without importing the reference:

  make_beta_schedule("linear")        ldm/modules/diffusionmodules/util.py:21-25
  DDPM.register_schedule              ldm/models/diffusion/ddpm.py:149-169
  make_ddim_timesteps("uniform")      util.py:46-60
  make_ddim_sampling_parameters       util.py:63-74
  DDIMSampler.make_schedule           ldm/models/diffusion/ddim.py:23-52
  DDIMSampler.ddim_sampling           ddim.py:224-302
  DDIMSampler.p_sample_ddim           ddim.py:304-386  (eps-parameterisation, CFG uncond-first)
  DiffusionWrapper.forward "hybrid"   ddpm.py:1348-1351
"""
import numpy as np
import torch


def alphas_cumprod(linear_start=0.00085, linear_end=0.0120, timesteps=1000):
    """float64 cumprod, then rounded to fp32 as the reference's registered buffer is (ddpm.py:166-169)."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas, axis=0)
    return ac.astype(np.float32)


def ddim_timesteps(S, T=1000):
    c = T // S
    return np.asarray(list(range(0, T, c))) + 1


def ddim_tables(S, eta, ac=None):
    """Returns dict of float64 tables of length len(timesteps): alphas, alphas_prev, sigmas, sqrt_one_minus_alphas.

    The reference indexes the fp32-rounded `alphas_cumprod` buffer (moved to CPU) with numpy, so table math runs
    in float32->float64 promotion exactly as numpy does there: alphacums is a float32 *torch* tensor indexed by a
    numpy array -> float32 tensor; `np.asarray([..] + tolist())` makes float64 from python floats.
    """
    ac = alphas_cumprod() if ac is None else ac
    ts = ddim_timesteps(S, ac.shape[0])
    a32 = torch.from_numpy(ac)
    alphas = a32[ts]                                               # float32 torch tensor
    alphas_prev = np.asarray([a32[0]] + a32[ts[:-1]].tolist())     # float64 numpy
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return {
        "timesteps": ts,
        "alphas": np.asarray(alphas, dtype=np.float64),
        "alphas_prev": np.asarray(alphas_prev, dtype=np.float64),
        "sigmas": np.asarray(sigmas, dtype=np.float64),
        "sqrt_one_minus_alphas": np.asarray(np.sqrt(1.0 - alphas), dtype=np.float64),
    }


def cfg_ddim_update(x, e_u, e_c, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """ddim.py:343-381 with fp32 state. Coefficients are cast to fp32 first (torch.full default dtype)."""
    f = lambda v: torch.tensor(float(np.float32(v)), dtype=torch.float32)
    a_t, a_prev, sigma_t, s1 = f(a_t), f(a_prev), f(sigma_t), f(sqrt_one_minus_at)
    e_t = e_u + scale * (e_c - e_u)
    e_t = e_t.float()
    pred_x0 = (x - s1 * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma_t * noise
    return x_prev, pred_x0


@torch.no_grad()
def ddim_sample(apply_unet, S, x_T, c_concat, c_cross, uc_cross, scale, eta=0.0, noises=None, log_every_t=100,
                trace=None):
    """DDIMSampler.sample -> ddim_sampling (ddim.py:224-302) for the hybrid-conditioned inpainting model.

    apply_unet(xc [2B,9,h,w], t [2B] int64, ctx [2B,L,D]) -> eps [2B,4,h,w]
    noises: optional list of S tensors (used when eta > 0; the reference draws randn every step, ddim.py:378).
    trace : optional list receiving (i, step, index) tuples -- pins the step indexing.
    """
    tabs = ddim_tables(S, eta)
    ts = tabs["timesteps"]
    img = x_T.clone().float()
    B = img.shape[0]
    inter = {"x_inter": [img], "pred_x0": [img]}
    time_range = np.flip(ts)
    total = ts.shape[0]
    for i, step in enumerate(time_range):
        index = total - i - 1
        if trace is not None:
            trace.append((i, int(step), index))
        t = torch.full((B,), int(step), dtype=torch.long)
        if scale == 1.0 or uc_cross is None:
            xc = torch.cat([img, c_concat], dim=1)
            e = apply_unet(xc, t, c_cross)
            e_u = e_c = e
        else:
            # uncond FIRST (ddim.py:317-333); c_concat is shared by both halves (ref_inpainting_ldm.py:50-51)
            xc = torch.cat([torch.cat([img] * 2), torch.cat([c_concat] * 2)], dim=1)
            e = apply_unet(xc, torch.cat([t] * 2), torch.cat([uc_cross, c_cross]))
            e_u, e_c = e.chunk(2)
        noise = None
        if noises is not None:
            noise = noises[i]
        img, pred_x0 = cfg_ddim_update(img, e_u, e_c, scale, tabs["alphas"][index], tabs["alphas_prev"][index],
                                       tabs["sigmas"][index], tabs["sqrt_one_minus_alphas"][index], noise)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


def multi_pick(n, rng=None):
    """Which condition's right half is shared after a step (ddim.py:208-212): `random.shuffle(list); list[0]`.
    Shuffling a list of indices consumes the python RNG exactly like shuffling the n tensors and yields the same choice."""
    import random
    order = list(range(n))
    (rng or random).shuffle(order)
    return order[0]


@torch.no_grad()
def ddim_multi_sample(apply_unet, S, x_Ts, c_concats, c_crosses, uc_crosses, scale, eta=0.0, noises=None, rng=None):
    """DDIMSampler.ddim_multi_sampling (ddim.py:147-222): K conditionings advance side by side; after every step the
    right half (last dim) of ONE randomly chosen condition overwrites the right half of all K states.  Returns img[0].

    noises: optional list of S*K tensors in call order (step-major, condition-minor), cf. ddim.py:378.
    """
    K = len(c_concats)
    tabs = ddim_tables(S, eta)
    ts = tabs["timesteps"]
    imgs = [x.clone().float() for x in x_Ts]
    B = imgs[0].shape[0]
    total = ts.shape[0]
    n = 0
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((B,), int(step), dtype=torch.long)
        new = []
        for k in range(K):
            xc = torch.cat([torch.cat([imgs[k]] * 2), torch.cat([c_concats[k]] * 2)], dim=1)
            e_u, e_c = apply_unet(xc, torch.cat([t] * 2), torch.cat([uc_crosses[k], c_crosses[k]])).chunk(2)
            noise = noises[n] if noises is not None else None
            n += 1
            x_prev, _ = cfg_ddim_update(imgs[k], e_u, e_c, scale, tabs["alphas"][index], tabs["alphas_prev"][index],
                                        tabs["sigmas"][index], tabs["sqrt_one_minus_alphas"][index], noise)
            new.append(x_prev)
        pick = multi_pick(K, rng)
        half = new[0].shape[-1] // 2
        right = new[pick][..., half:].clone()
        for k in range(K):
            new[k][..., half:] = right
        imgs = new
    return imgs[0]


def p_losses(apply_unet, x_start, c_concat, c_cross, t, noise, ac=None):
    """LatentDiffusion.p_losses (ddpm.py:900-935) for the eps-parameterised hybrid inpainting model with the shipped
    settings (l2 loss, logvar = 0 buffer, l_simple_weight 1, original_elbo_weight 0): q_sample -> UNet -> MSE.

    apply_unet(xc [B,9,h,w], t, ctx) -> eps (any float dtype).  Returns (loss, loss_simple, x_noisy); differentiable
    w.r.t. c_cross when apply_unet is (test infrastructure: the checker for the HIP backward)."""
    ac = torch.from_numpy(alphas_cumprod() if ac is None else ac).float()
    shape = (-1, 1, 1, 1)
    x_noisy = ac.sqrt()[t].reshape(shape) * x_start + (1.0 - ac).sqrt()[t].reshape(shape) * noise      # ddpm.py:370-373
    eps = apply_unet(torch.cat([x_noisy, c_concat], dim=1), t, c_cross).float()
    loss_simple = torch.nn.functional.mse_loss(noise, eps, reduction="none").mean([1, 2, 3])
    return loss_simple.mean(), loss_simple, x_noisy
