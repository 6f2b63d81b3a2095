"""`inpainting_ldm.ref_inpainting_ldm.RefInpaintLDM` -- the inference entry points of LeftRefill on the MI355X build.

Reproduces `log_images` / `sample_log` / `get_unconditional_conditioning` (reference
inpainting_ldm/ref_inpainting_ldm.py:30-81) and the attributes callers read (`cond_cfg`, `data_cfg`, `world_size`,
`loss_fn_alex`, `save_prompt_only`, test_inpainting.py:95-110).  The Lightning training / validation hooks
(83-173) are out of scope.
"""
import torch

from ldm.models.diffusion.ddim import DDIMSampler
from ldm.models.diffusion.ddpm import LatentInpaintDiffusion


class RefInpaintLDM(LatentInpaintDiffusion):
    def __init__(self, *args, **kwargs):
        data_cfg = kwargs.pop('data_config', None)
        save_prompt_only = kwargs.pop('save_prompt_only', False)
        cond_cfg = kwargs.get('cond_stage_config')
        super().__init__(*args, **kwargs)
        self.loss_fn_alex = None
        self.cfg = None
        self.optim_cfg = None
        self.data_cfg = dict(data_cfg) if data_cfg is not None else {}
        self.cond_cfg = dict(cond_cfg.get('params', {}) or {}) if isinstance(cond_cfg, dict) else {}
        self.world_size = 1
        self.image_text_pair = False
        self.img_size = self.data_cfg.pop('img_size', 256)
        self.save_prompt_only = save_prompt_only

    @torch.no_grad()
    def get_unconditional_conditioning(self, N):
        if self.cond_cfg.get('deep_prompt', False):
            return self.get_learned_conditioning([[""] * N] * self.cond_cfg['cross_attn_layers'])
        return self.get_learned_conditioning([""] * N)

    @torch.no_grad()
    def log_images(self, batch, N=4, ddim_steps=50, ddim_eta=0.0, unconditional_guidance_scale=9.0, **kwargs):
        use_ddim = ddim_steps is not None
        log = dict()
        z, c = self.get_input(batch, self.first_stage_key, bs=N)
        c_concat, c_crossattn = c["c_concat"][0][:N], c["c_crossattn"][0][:N]
        N = min(z.shape[0], N)
        log["masked_image"] = batch['masked_image'].permute(0, 3, 1, 2)
        log["origin_image"] = batch['image'].permute(0, 3, 1, 2)
        if unconditional_guidance_scale > 1.0:
            uc_full = {"c_concat": [c_concat], "c_crossattn": [self.get_unconditional_conditioning(N)]}
            samples, _ = self.sample_log(cond={"c_concat": [c_concat], "c_crossattn": [c_crossattn]}, batch_size=N,
                                         ddim=use_ddim, ddim_steps=ddim_steps, eta=ddim_eta,
                                         unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=uc_full)
        elif unconditional_guidance_scale == 0.0:
            uc_cross = self.get_unconditional_conditioning(N)
            samples, _ = self.sample_log(cond={"c_concat": [c_concat], "c_crossattn": [uc_cross]}, batch_size=N,
                                         ddim=use_ddim, ddim_steps=ddim_steps, eta=ddim_eta)
        else:
            samples, _ = self.sample_log(cond={"c_concat": [c_concat], "c_crossattn": [c_crossattn]}, batch_size=N,
                                         ddim=use_ddim, ddim_steps=ddim_steps, eta=ddim_eta)
        log["pred"] = self.decode_first_stage(samples)
        return log

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        sampler = DDIMSampler(self)
        _, _, h, w = cond["c_concat"][0].shape
        shape = (self.channels, h, w)   # latent size comes from c_concat (reference 77-79)
        return sampler.sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)
