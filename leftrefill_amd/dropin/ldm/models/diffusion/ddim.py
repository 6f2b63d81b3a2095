"""`ldm.models.diffusion.ddim.DDIMSampler` for the MI355X build (sampling only).

Same constructor / `sample` / `ddim_sampling` / `p_sample_ddim` / `make_schedule` surface as the reference
(ldm/models/diffusion/ddim.py:11-386) for the paths LeftRefill uses (eps-parameterisation, uniform discretisation,
dict conditioning, classifier-free guidance with the unconditional batch FIRST).

Differences by design (results identical):
  * schedule tables stay on the HOST as float64/fp32 numpy -- the reference builds four `torch.full(...)` from 0-dim
    device tensors per step (ddim.py:359-362), i.e. four device->host syncs per step;
  * the CFG combine + x0 prediction + x_{t-1} update (ddim.py:343-381, ~15 elementwise kernels) is ONE HIP kernel
    (lr_ddim_cfg_step); the UNet step itself is one hipGraph replay.
"""
import numpy as np
import torch

from leftrefill_amd import ops
from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps, noise_like


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps,
                                                  verbose=verbose)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        ac32 = ac.detach().to(torch.float32).cpu().numpy()
        self.alphas_cumprod = ac32
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac32, self.ddim_timesteps, ddim_eta, verbose=verbose)
        self.ddim_sigmas = sigmas
        self.ddim_alphas = alphas
        self.ddim_alphas_prev = alphas_prev
        # sqrt(1 - a_t) is formed in fp32 by the reference (np.sqrt of an fp32 tensor, ddim.py:47)
        a32 = ac32[self.ddim_timesteps]
        self.ddim_sqrt_one_minus_alphas = np.sqrt(np.float32(1) - a32).astype(np.float64)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, **kwargs):
        if conditioning is not None and isinstance(conditioning, dict):
            c0 = conditioning[list(conditioning.keys())[0]]
            while isinstance(c0, list):
                c0 = c0[0]
            if c0.shape[0] != batch_size:
                print(f"Warning: Got {c0.shape[0]} conditionings but batch-size is {batch_size}")
        if quantize_x0 or score_corrector is not None or dynamic_threshold is not None or noise_dropout > 0.:
            raise NotImplementedError("quantize_x0 / score_corrector / dynamic_threshold / noise_dropout are unused")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        if isinstance(conditioning, list):       # NVS consistency sampler (ddim.py:103-120)
            return self.ddim_multi_sampling(conditioning, (batch_size, C, H, W), callback=callback, temperature=temperature,
                                            x_T=x_T, unconditional_guidance_scale=unconditional_guidance_scale,
                                            unconditional_conditioning=unconditional_conditioning,
                                            ucg_schedule=ucg_schedule)
        return self.ddim_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback,
                                  mask=mask, x0=x0, temperature=temperature, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, ucg_schedule=ucg_schedule)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, timesteps=None, mask=None, x0=None,
                      img_callback=None, log_every_t=100, temperature=1., unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, ucg_schedule=None, **kwargs):
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        steps = self.ddim_timesteps
        if timesteps is not None:
            end = int(min(timesteps / steps.shape[0], 1) * steps.shape[0]) - 1
            steps = steps[:end]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = np.flip(steps)
        total_steps = steps.shape[0]
        self._prepare_cfg_inputs(cond, unconditional_conditioning, unconditional_guidance_scale)
        self._prepare_timesteps(time_range)
        for i, step in enumerate(time_range):
            index = total_steps - i - 1          # bit-identical step indexing (ddim.py:254)
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img = self.model.q_sample(x0, ts) * mask + (1. - mask) * img
            if ucg_schedule is not None:
                unconditional_guidance_scale = ucg_schedule[i]
            # t_host: the timestep of `ts` as a host integer -- p_sample_ddim names it to the UNet around its own model calls
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, t_host=int(step))
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        self._cfg_cache = None
        return img, intermediates

    @torch.no_grad()
    def ddim_multi_sampling(self, cond, shape, x_T=None, callback=None, timesteps=None, temperature=1.,
                            unconditional_guidance_scale=1., unconditional_conditioning=None, ucg_schedule=None,
                            **kwargs):
        """K conditionings denoised side by side; after every step the right half of ONE of them -- chosen with python's
        `random.shuffle`, consuming the RNG exactly like the reference -- replaces the right half of all K states
        (reference ddim.py:147-222).  Returns (img[0], {})."""
        import random
        device = self.model.betas.device
        b = shape[0]
        K = len(cond)
        if x_T is None:
            first = torch.randn(shape, device=device)       # the reference aliases ONE randn tensor K times (ddim.py:160)
            img = [first] * K
        else:
            img = [x.to(device=device, dtype=torch.float32) for x in x_T]
        ucs = unconditional_conditioning if unconditional_conditioning is not None else [None] * K
        steps = self.ddim_timesteps
        if timesteps is not None:
            end = int(min(timesteps / steps.shape[0], 1) * steps.shape[0]) - 1
            steps = steps[:end]
        total_steps = steps.shape[0]
        self._prepare_timesteps(steps)
        for i, step in enumerate(np.flip(steps)):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if ucg_schedule is not None:
                unconditional_guidance_scale = ucg_schedule[i]
            new_img = []
            for img_, cond_, uc_ in zip(img, cond, ucs):
                x_prev, _ = self.p_sample_ddim(img_, cond_, ts, index=index, temperature=temperature,
                                               unconditional_guidance_scale=unconditional_guidance_scale,
                                               unconditional_conditioning=uc_, t_host=int(step))
                new_img.append(x_prev)
            order = list(range(K))
            random.shuffle(order)                 # same RNG consumption and same pick as shuffling the K tensors
            half = new_img[0].shape[-1] // 2
            right = new_img[order[0]][..., half:].clone()
            for x_ in new_img:
                x_[..., half:] = right
            img = new_img
            if callback:
                callback(i)
        return img[0], {}

    def _unet(self):
        return getattr(getattr(self.model, "model", None), "diffusion_model", None)

    def _prepare_timesteps(self, steps):
        """The schedule is known before the first step: the UNet computes the embedding rows of all its timesteps in one go
        (UNetModel.prepare_timesteps) and each step names its timestep on the host (`_step_hint`)."""
        unet = self._unet()
        if hasattr(unet, "prepare_timesteps"):
            unet.prepare_timesteps(int(s_) for s_ in steps)

    def _step_hint(self, step):
        unet = self._unet()
        if hasattr(unet, "prepare_timesteps"):
            unet._t_host = None if step is None else int(step)

    # the conditioning is constant over the loop: build the [uncond; cond] batch once instead of 50 torch.cat calls
    def _prepare_cfg_inputs(self, c, uc, scale):
        self._cfg_cache = None
        self._cfg_shared = False
        if uc is None or scale == 1. or not isinstance(c, dict):
            return
        c_in = {}
        for k in c:
            if isinstance(c[k], list):
                c_in[k] = [torch.cat([uc[k][i], c[k][i]]) for i in range(len(c[k]))]
            else:
                c_in[k] = torch.cat([uc[k], c[k]])
        self._cfg_cache = (id(c), id(uc), c_in)
        # the two halves of the CFG batch differ only in the cross-attention context when every other conditioning tensor
        # is the same for uncond and cond: checked once per sampling, lets the UNet share the context-free prefix
        import os
        self._cfg_shared = os.environ.get("LEFTREFILL_CFG_SHARED_PREFIX", "1") != "0" and all(
            all(torch.equal(u_, c_) for u_, c_ in zip(uc[k], c[k])) if isinstance(c[k], list) else torch.equal(uc[k], c[k])
            for k in c if k != "c_crossattn")

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
                      t_host=None, **kwargs):
        """t_host: the timestep every entry of `t` holds, as a host integer (the sampling loops pass it; None = unknown).  It is named
        to the UNet only around this method's own apply_model calls (precomputed embedding rows, UNetModel.prepare_timesteps)."""
        self._step_hint(t_host)
        try:
            return self._p_sample_ddim(x, c, t, index, repeat_noise, use_original_steps, quantize_denoised, temperature, noise_dropout,
                                       score_corrector, corrector_kwargs, unconditional_guidance_scale, unconditional_conditioning,
                                       dynamic_threshold)
        finally:
            self._step_hint(None)

    def _p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                       temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                       unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None):
        if use_original_steps or quantize_denoised or score_corrector is not None or dynamic_threshold is not None:
            raise NotImplementedError
        if self.model.parameterization != "eps":
            raise NotImplementedError("LeftRefill samples in the eps-parameterisation")
        x = x.float().contiguous()
        device = x.device
        scale = float(unconditional_guidance_scale)
        if unconditional_conditioning is None or scale == 1.:
            e = self.model.apply_model(x, t, c)
            eps = torch.cat([e, e])      # degenerate CFG: e_u = e_c = e  ->  e_t = e
            scale = 1.0
        else:
            from leftrefill_amd import dist as lrd
            if lrd.split_cfg_active():
                # cond / uncond passes on two ranks (or, without a process group, one after the other): batch B each, one
                # all-gather of the eps halves per step -- leftrefill_amd/dist.py.  (The [uncond; cond] batch is not built here.)
                # The ranks of a pair must hold the same x and draw the same DDIM noise: the caller seeds them identically
                # (bench.py: seed 1234 + rank // 2).
                role = lrd.split_cfg_role()
                if role is None:
                    # one process runs both passes: each gets its own captured step (graph slot), so the per-context K / V cache
                    # of a slot sees ONE context and is not recomputed twice per DDIM step
                    unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
                    halves = []
                    for slot, cc in enumerate((unconditional_conditioning, c)):
                        if unet is not None:
                            unet._graph_slot = slot
                        try:
                            halves.append(self.model.apply_model(x, t, cc))
                        finally:
                            if unet is not None:
                                unet._graph_slot = 0
                    eps = torch.cat(halves)
                else:
                    eps = lrd.cfg_exchange(self.model.apply_model(x, t, unconditional_conditioning if role == 0 else c))
                noise = noise_like(x.shape, device, repeat_noise)
                sigma = float(self.ddim_sigmas[index])
                return ops.ddim_cfg_step(x, eps.contiguous(), noise, scale, self.ddim_alphas[index], self.ddim_alphas_prev[index],
                                         sigma * float(temperature), self.ddim_sqrt_one_minus_alphas[index])
            cache = getattr(self, "_cfg_cache", None)
            if cache is not None and cache[0] == id(c) and cache[1] == id(unconditional_conditioning):
                c_in = cache[2]
            else:
                assert isinstance(c, dict) and isinstance(unconditional_conditioning, dict)
                c_in = {k: ([torch.cat([unconditional_conditioning[k][i], c[k][i]]) for i in range(len(c[k]))]
                            if isinstance(c[k], list) else torch.cat([unconditional_conditioning[k], c[k]]))
                        for k in c}
            x_in = torch.cat([x] * 2)
            t_in = torch.cat([t] * 2)
            unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
            shared = (cache is not None and c_in is cache[2] and getattr(self, "_cfg_shared", False)
                      and hasattr(unet, "cfg_shared_prefix"))
            if shared:
                unet.cfg_shared_prefix = True
            try:
                eps = self.model.apply_model(x_in, t_in, c_in)   # [2B, 4, h, w], uncond half first (ddim.py:317-342)
            finally:
                if shared:
                    unet.cfg_shared_prefix = False
        eps = eps.contiguous()
        # randn is drawn every step like the reference (ddim.py:378), also when sigma_t == 0
        noise = noise_like(x.shape, device, repeat_noise)
        sigma = float(self.ddim_sigmas[index])
        x_prev, pred_x0 = ops.ddim_cfg_step(x, eps, noise, scale, self.ddim_alphas[index], self.ddim_alphas_prev[index],
                                            sigma * float(temperature), self.ddim_sqrt_one_minus_alphas[index])
        return x_prev, pred_x0
