"""`inpainting_ldm.NVS_ldm` -- the novel-view-synthesis task model (reference inpainting_ldm/NVS_ldm.py; BASELINE configs[4] trains it:
configs/nvs_training_config.yaml) on the MI355X build.

`NVSUnetModel` (reference 22-104) is the UNet with two input-side additions, both index arithmetic around the same HIP blocks:
  * `c_input` (64-68): a tensor added to the output of the first input block (whole canvas, or only its right half) -- the output
    of the optional input-refinement network;
  * `use_sep` (25-32, 57-60, 70-99): a learned separator column `sep_token[str(C)]` is inserted between the left and the right half
    before every block that does not end in a Down / Upsample (canvas width W -> W + 1) and cut out again after it.
`NVSLDM` (107-298): conditioning glue (get_input with the refinement branch, unconditional prompts incl. deep prompts), `log_images`,
`log_multi_cond_images` (K conditionings through DDIMSampler.ddim_multi_sampling) and `sample_log`; the parameter groups its
optimizer owns (`trainable_parameters`, reference 314-337).  The Lightning hooks, dataloaders, LoRA injection and validation metrics
(299-435) stay out of scope.  The shipped configs use the plain UNetModel with `use_sep: False` and no refinement; the
class is kept a drop-in for checkpoints / configs that turn them on.
"""
import torch
import torch.nn as nn

from leftrefill_amd import engine
from ldm.models.diffusion.ddim import DDIMSampler
from ldm.models.diffusion.ddpm import LatentInpaintDiffusion
from ldm.modules.diffusionmodules.openaimodel import UNetModel
from ldm.modules.diffusionmodules.util import GroupNorm32, conv_nd

SEP_CHANNELS = (9, 320, 640, 1280, 2560, 1920, 960)      # widths a block input can have in the SD2 UNet (reference 27)


class NVSUnetModel(UNetModel):
    def __init__(self, *args, **kwargs):
        self.use_sep = kwargs.pop("use_sep", False)
        super().__init__(*args, **kwargs)
        self.sep_token = None
        if self.use_sep:
            self.sep_token = nn.ParameterDict({str(ch): nn.Parameter(torch.randn(ch), requires_grad=True) for ch in SEP_CHANNELS})
            self.eager_only = True        # block shapes change inside the step: no captured graph for this variant

    def _needs_autograd(self, context, c_input=None):
        # the separator tokens are trainable parameters of the UNet itself: their gradient needs the autograd path too
        sep_grad = self.use_sep and any(p.requires_grad for p in self.sep_token.values())
        return super()._needs_autograd(context, c_input) or (torch.is_grad_enabled() and sep_grad)

    def _block_in(self, act, steps):
        """[left | right] -> [left | sep | right] for blocks that do not end in a Down / Upsample (57-60, 85-88)."""
        if not self.use_sep or steps[-1][0] in ("down", "up"):
            return act, None
        N, H, W = act.N, act.H, act.W
        tok = act.materialize().reshape(N, H, W, -1)
        C = tok.shape[-1]
        key = str(self.in_channels) if C not in SEP_CHANNELS and str(self.in_channels) in self.sep_token else str(C)
        sep = self.sep_token[key].to(tok.dtype)
        if sep.numel() < C:               # the 9 input channels are zero-padded to the kernels' 64
            sep = torch.cat([sep, sep.new_zeros(C - sep.numel())])
        col = sep.reshape(1, 1, 1, C).expand(N, H, 1, C)
        tok = torch.cat([tok[:, :, :W // 2], col, tok[:, :, W // 2:]], dim=2)
        return engine.Act(tok.reshape(N * H * (W + 1), C).contiguous(), N, H, W + 1), W

    def _block_out(self, act, W):
        if W is None:
            return act
        N, H = act.N, act.H
        tok = act.materialize().reshape(N, H, act.W, -1)
        tok = torch.cat([tok[:, :, :W // 2], tok[:, :, act.W - W // 2:]], dim=2)
        return engine.Act(tok.reshape(N * H * W, -1).contiguous(), N, H, W)


def refinement_network(model_channels):
    """Input-refinement side network (reference 124-143): masked image + sub-pixel mask at image resolution -> a model_channels map
    at latent resolution; a handful of small convolutions per batch, run once per sampling (host PyTorch code like the VAE glue)."""
    return nn.Sequential(conv_nd(2, 4, 32, 3, padding=1), nn.SiLU(),
                         conv_nd(2, 32, 64, 3, padding=1, stride=2), GroupNorm32(16, 64), nn.SiLU(),
                         conv_nd(2, 64, 64, 3, padding=1), GroupNorm32(16, 64), nn.SiLU(),
                         conv_nd(2, 64, 128, 3, padding=1, stride=2), GroupNorm32(32, 128), nn.SiLU(),
                         conv_nd(2, 128, 128, 3, padding=1), GroupNorm32(32, 128), nn.SiLU(),
                         conv_nd(2, 128, 256, 3, padding=1, stride=2), GroupNorm32(32, 256), nn.SiLU(),
                         conv_nd(2, 256, model_channels, 3, padding=1), GroupNorm32(32, model_channels), nn.SiLU())


class NVSLDM(LatentInpaintDiffusion):
    def __init__(self, *args, **kwargs):
        data_cfg = kwargs.pop("data_config", None)
        self_save_prompt_only = kwargs.pop("save_prompt_only", False)
        refinement = kwargs.pop("refinement_config", None) or {"use_input_refinement": False, "only_masked_refine": False}
        lora = kwargs.pop("lora", None) or {"do_lora": False}
        cond_cfg = kwargs.get("cond_stage_config")
        super().__init__(*args, **kwargs)
        if lora.get("do_lora"):
            raise NotImplementedError("LoRA injection (inpainting_ldm/lora.py) is out of scope of this build")
        self.loss_fn_alex = None
        self.cfg = None
        self.optim_cfg = None
        self.data_cfg = dict(data_cfg) if data_cfg is not None else {}
        self.cond_cfg = dict(cond_cfg.get("params", {}) or {}) if isinstance(cond_cfg, dict) else {}
        self.world_size = 1
        self.image_text_pair = False
        self.img_size = self.data_cfg.pop("img_size", 256)
        self.mask_steps = 0
        self.warmup_mask_steps = self.data_cfg.get("warmup_mask_steps", 0)
        self.complete_mask_rate = self.data_cfg.get("complete_mask_rate", 0)
        self.save_prompt_only = self_save_prompt_only
        self.refinement_config = dict(refinement)
        self.lora_cfg = dict(lora)
        self.unet_lora_params = None
        self.refinement_model = self.refinement_alpha = None
        if self.refinement_config.get("use_input_refinement"):
            self.refinement_model = refinement_network(kwargs["unet_config"]["params"]["model_channels"])
            self.refinement_alpha = nn.Parameter(torch.tensor(0, dtype=torch.float32), requires_grad=True)

    def get_input(self, batch, k, cond_key=None, bs=None, return_first_stage_outputs=False, force_c_encode=True):
        x, c = super().get_input(batch, k, cond_key, bs, return_first_stage_outputs, force_c_encode)
        c["c_input"] = None
        if self.refinement_config.get("use_input_refinement"):
            masked_key, mask_key = (("clean_masked_image", "clean_mask") if self.refinement_config.get("only_masked_refine", False)
                                    else ("masked_image", "subpixel_mask"))
            to_nchw = lambda t: t.permute(0, 3, 1, 2).to(self.device).contiguous().float()      # 'b h w c -> b c h w'
            inp = torch.cat([to_nchw(batch[masked_key]), to_nchw(batch[mask_key])], dim=1)
            if bs is not None:
                inp = inp[:bs]
            c["c_input"] = self.refinement_model(inp) * self.refinement_alpha
        return x, c

    @torch.no_grad()
    def get_unconditional_conditioning(self, N):
        if self.cond_cfg.get("deep_prompt", False):
            return self.get_learned_conditioning([[""] * N] * self.cond_cfg["cross_attn_layers"])
        return self.get_learned_conditioning([""] * N)

    def _cond_pair(self, batch, N):
        """(cond, uncond) dicts of one batch for classifier-free guidance: same c_concat / c_input, empty prompts (247-262)."""
        z, c = self.get_input(batch, self.first_stage_key, bs=N)
        n = min(z.shape[0], N)
        c_full = {"c_concat": [c["c_concat"][0][:N]], "c_crossattn": [c["c_crossattn"][0][:N]]}
        uc_full = {"c_concat": [c_full["c_concat"][0]], "c_crossattn": [self.get_unconditional_conditioning(n)]}
        if c.get("c_input") is not None:
            c_full["c_input"] = c["c_input"][:N]
            uc_full["c_input"] = c_full["c_input"].clone()
        return n, c_full, uc_full

    @torch.no_grad()
    def log_images(self, batch, N=4, ddim_steps=50, ddim_eta=0.0, unconditional_guidance_scale=9.0, **kwargs):
        log = {"masked_image": batch["masked_image"].permute(0, 3, 1, 2), "origin_image": batch["image"].permute(0, 3, 1, 2)}
        n, c_full, uc_full = self._cond_pair(batch, N)
        if unconditional_guidance_scale > 1.0:
            samples, _ = self.sample_log(cond=c_full, batch_size=n, ddim=ddim_steps is not None, ddim_steps=ddim_steps, eta=ddim_eta,
                                         unconditional_guidance_scale=unconditional_guidance_scale, unconditional_conditioning=uc_full)
        else:      # (the reference drops c_input on this branch, 274-276)
            samples, _ = self.sample_log(cond={k: c_full[k] for k in ("c_concat", "c_crossattn")}, batch_size=n,
                                         ddim=ddim_steps is not None, ddim_steps=ddim_steps, eta=ddim_eta)
        log["pred"] = self.decode_first_stage(samples)
        return log

    @torch.no_grad()
    def log_multi_cond_images(self, batch, N=4, ddim_steps=50, ddim_eta=0.0, unconditional_guidance_scale=9.0, **kwargs):
        """`batch` = list of K batches (K reference views of the same targets): the K conditionings are denoised side by side and
        share their right half after every step (DDIMSampler.ddim_multi_sampling; reference 283-319)."""
        assert isinstance(batch, list)
        log = {"masked_image": batch[0]["masked_image"].permute(0, 3, 1, 2), "origin_image": batch[0]["image"].permute(0, 3, 1, 2)}
        conds, unconds, n = [], [], N
        for b_ in batch:
            n, c_full, uc_full = self._cond_pair(b_, N)
            conds.append(c_full)
            unconds.append(uc_full)
        samples, _ = self.sample_log(cond=conds, batch_size=n, ddim=ddim_steps is not None, ddim_steps=ddim_steps, eta=ddim_eta,
                                     unconditional_guidance_scale=unconditional_guidance_scale, unconditional_conditioning=unconds)
        log["pred"] = self.decode_first_stage(samples)
        return log

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        first = cond[0] if isinstance(cond, list) else cond
        _, _, h, w = first["c_concat"][0].shape
        return DDIMSampler(self).sample(ddim_steps, batch_size, (self.channels, h, w), cond, verbose=False, **kwargs)

    def trainable_parameters(self):
        """What the reference's optimizer owns besides an optional full-UNet group (configure_optimizers, 314-337): the learned
        prompt tokens, the pose MLP, the refinement network and the separator tokens' owner decides on those itself."""
        params = list(self.cond_stage_model.special_embeddings.parameters())
        if getattr(self.cond_stage_model, "rel_pos_model", None) is not None:
            params.extend(self.cond_stage_model.rel_pos_model.parameters())
        if self.refinement_model is not None and self.refinement_alpha is not None:
            params.extend(self.refinement_model.parameters())
            params.append(self.refinement_alpha)
        return params
