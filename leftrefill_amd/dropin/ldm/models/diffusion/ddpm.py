"""`ldm.models.diffusion.ddpm` -- inference-side host glue for the MI355X build.

Keeps the reference's class names, constructor kwargs and attributes that the sampler / callers read
(reference ldm/models/diffusion/ddpm.py: DDPM 46-535, LatentDiffusion 538-1324, DiffusionWrapper 1327-1371,
LatentFinetuneDiffusion 1512-1651, LatentInpaintDiffusion 1654-1701), minus the PyTorch-Lightning trainer hooks,
EMA, logging and the unused upscale/depth variants (out of scope).

On the hot path only `apply_model` -> `DiffusionWrapper.forward` ('hybrid': channel-concat of the noisy latent with
[mask | masked-image latent], cross-attention context) -> `UNetModel.forward` is exercised; VAE encode/decode and the
prompt encoder stay PyTorch-ROCm host code as the north star prescribes.
"""
import numpy as np
import torch
import torch.nn as nn

from ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
from ldm.util import default, exists, instantiate_from_config


def disabled_train(self, mode=True):
    return self


class DiffusionWrapper(nn.Module):
    """Conditioning router (reference 1327-1371).  state-dict prefix: `model.diffusion_model.*`."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.sequential_cross_attn = diff_model_config.pop("sequential_crossattn", False)
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, 'concat', 'crossattn', 'hybrid', 'adm', 'hybrid-adm', 'crossattn-adm',
                                         'hybrid-refine']

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, c_adm=None, c_input=None):
        key = self.conditioning_key
        if key is None:
            raise NotImplementedError("unconditional UNet is not a LeftRefill configuration")
        if key == 'concat':
            raise NotImplementedError("LeftRefill's UNet always has a cross-attention context")
        if key == 'crossattn':
            cc = torch.cat(c_crossattn, 1)
            return self.diffusion_model(x, t, context=cc)
        if key == 'hybrid':
            # channel order [noisy z 0-3 | mask 4 | masked-image latent 5-8] (reference 1348-1351, 1662, 1679-1690)
            xc = torch.cat([x] + c_concat, dim=1)
            # a single context tensor is passed through as-is (torch.cat would copy it every step and defeat the
            # UNet's per-context K/V cache); the value is identical
            cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
            if c_input is not None:      # NVS input refinement (reference 1355, inpainting_ldm/NVS_ldm.py:64-68)
                return self.diffusion_model(xc, t, context=cc, c_input=c_input)
            return self.diffusion_model(xc, t, context=cc)
        raise NotImplementedError(f"conditioning_key {key!r} is not used by the inpainting path")


class DDPM(nn.Module):
    """Noise schedule buffers (reference register_schedule 149-203) + the attributes DDIMSampler reads."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=(), load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1.,
                 conditioning_key=None, parameterization="eps", scheduler_config=None, use_positional_encodings=False,
                 learn_logvar=False, logvar_init=0., make_it_fit=False, ucg_training=None, reset_ema=False,
                 reset_num_ema_updates=False, **ignored):
        super().__init__()
        assert parameterization in ["eps", "x0", "v"]
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.first_stage_key = first_stage_key
        self.image_size = image_size
        self.channels = channels
        self.use_positional_encodings = use_positional_encodings
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = False  # every LeftRefill config sets use_ema: False
        self.v_posterior = v_posterior
        self.original_elbo_weight = original_elbo_weight
        self.l_simple_weight = l_simple_weight
        self.loss_type = loss_type
        self.learn_logvar = learn_logvar
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        logvar = torch.full(fill_value=float(logvar_init), size=(self.num_timesteps,))     # reference 129-134
        if learn_logvar:
            self.logvar = nn.Parameter(logvar, requires_grad=True)
        else:
            self.register_buffer('logvar', logvar)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if exists(given_betas) else make_beta_schedule(beta_schedule, timesteps, linear_start,
                                                                           linear_end, cosine_s)
        betas = np.asarray(betas, dtype=np.float64)
        ac = np.cumprod(1. - betas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        for name, val in (("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
                          ("sqrt_alphas_cumprod", np.sqrt(ac)), ("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - ac)),
                          ("log_one_minus_alphas_cumprod", np.log(1. - ac)),
                          ("sqrt_recip_alphas_cumprod", np.sqrt(1. / ac)),
                          ("sqrt_recipm1_alphas_cumprod", np.sqrt(1. / ac - 1))):
            self.register_buffer(name, f32(val))
        # variational-bound weights of the training loss (reference 176-203), eps-parameterisation
        alphas = 1. - betas
        post_var = (1 - self.v_posterior) * betas * (1. - ac_prev) / (1. - ac) + self.v_posterior * betas
        self.register_buffer("posterior_variance", f32(post_var))
        if self.parameterization == "eps":
            lvlb = self.betas ** 2 / (2 * self.posterior_variance * f32(alphas) * (1 - self.alphas_cumprod))
        elif self.parameterization == "x0":
            lvlb = 0.5 * torch.sqrt(f32(ac)) / (2. * 1 - f32(ac))
        else:
            lvlb = torch.ones_like(self.betas)
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", lvlb, persistent=False)

    def get_loss(self, pred, target, mean=True):
        """reference 378-391 (under the reference's autocast the fp16 prediction is promoted to fp32 here)"""
        pred = pred.to(target.dtype)
        if self.loss_type == 'l1':
            loss = (target - pred).abs()
            return loss.mean() if mean else loss
        if self.loss_type == 'l2':
            return torch.nn.functional.mse_loss(target, pred, reduction='mean' if mean else 'none')
        raise NotImplementedError(f"unknown loss type '{self.loss_type}'")

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def get_input(self, batch, k):
        x = batch[k]
        if x.dim() == 3:
            x = x[..., None]
        return x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()   # 'b h w c -> b c h w'


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image",
                 cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None, conditioning_key=None,
                 scale_factor=1.0, scale_by_std=False, force_null_conditioning=False, *args, **kwargs):
        self.force_null_conditioning = force_null_conditioning
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        self.scale_by_std = scale_by_std
        if conditioning_key is None:
            conditioning_key = 'concat' if concat_mode else 'crossattn'
        if cond_stage_config == '__is_unconditional__' and not force_null_conditioning:
            conditioning_key = None
        kwargs.pop("ckpt_path", None)
        kwargs.pop("ignore_keys", None)
        super().__init__(conditioning_key=conditioning_key, *args, **kwargs)
        self.concat_mode = concat_mode
        self.cond_stage_trainable = cond_stage_trainable
        self.cond_stage_key = cond_stage_key
        self.scale_factor = scale_factor
        self.cond_stage_forward = cond_stage_forward
        self.first_stage_model = instantiate_from_config(first_stage_config)
        if self.first_stage_model is not None:
            self.first_stage_model.eval()
            self.first_stage_model.train = disabled_train
            for p in self.first_stage_model.parameters():
                p.requires_grad = False
        if cond_stage_config == "__is_first_stage__":
            self.cond_stage_model = self.first_stage_model
        elif cond_stage_config == "__is_unconditional__":
            self.cond_stage_model = None
        else:
            self.cond_stage_model = instantiate_from_config(cond_stage_config)

    # ---- first stage (KL-VAE; PyTorch-ROCm host code) -----------------------------------------------------------
    def get_first_stage_encoding(self, encoder_posterior):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        return self.first_stage_model.decode(1. / self.scale_factor * z)

    def get_learned_conditioning(self, c):
        m = self.cond_stage_model
        if self.cond_stage_forward is None:
            if hasattr(m, 'encode') and callable(m.encode):
                c = m.encode(c)
                if isinstance(c, DiagonalGaussianDistribution):
                    c = c.mode()
            else:
                c = m(c)
        else:
            c = getattr(m, self.cond_stage_forward)(c)
        return c

    def get_input(self, batch, k, return_first_stage_outputs=False, force_c_encode=False, cond_key=None,
                  return_original_cond=False, bs=None, return_x=False):
        x = super().get_input(batch, k)
        if bs is not None:
            x = x[:bs]
        x = x.to(self.device)
        z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
        c, xc = None, None
        if self.model.conditioning_key is not None and not self.force_null_conditioning:
            cond_key = default(cond_key, self.cond_stage_key)
            if cond_key == self.first_stage_key:
                xc = x
            elif cond_key in ('caption', 'coordinates_bbox', 'txt'):
                xc = batch[cond_key]
            elif cond_key == "txt+rel_pose":
                xc = [batch['txt'], batch['rel_pose'].to(self.device)]
            else:
                xc = super().get_input(batch, cond_key).to(self.device)
            if not self.cond_stage_trainable or force_c_encode:
                c = self.get_learned_conditioning(xc if isinstance(xc, (dict, list)) else xc.to(self.device))
            else:
                c = xc
            if bs is not None:
                c = c[:bs]
        out = [z, c]
        if return_first_stage_outputs:
            out.extend([x, self.decode_first_stage(z)])
        if return_x:
            out.append(x)
        if return_original_cond:
            out.append(xc)
        return out

    # ---- the hot path ---------------------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """reference 865-880: dict cond (hybrid) is passed through; otherwise wrapped by the conditioning key."""
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            cond = {('c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn'): cond}
        out = self.model(x_noisy, t, **cond)
        if isinstance(out, tuple) and not return_ids:
            return out[0]
        return out


    # ---- training objective (reference 854-863, 900-935); the UNet backward runs on the HIP kernels (train_ops) ----------
    def forward(self, x, c, *args, **kwargs):
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
        assert self.model.conditioning_key is None or c is not None
        return self.p_losses(x, c, t, *args, **kwargs)

    def p_losses(self, x_start, cond, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        model_output = self.apply_model(x_noisy, t, cond)
        prefix = 'train' if self.training else 'val'
        if self.parameterization == "x0":
            target = x_start
        elif self.parameterization == "eps":
            target = noise
        else:
            raise NotImplementedError("v-parameterisation is not used by LeftRefill")
        loss_dict = {}
        loss_simple = self.get_loss(model_output, target, mean=False).mean([1, 2, 3])
        loss_dict[f'{prefix}/loss_simple'] = loss_simple.mean()
        logvar_t = self.logvar[t].to(self.device)
        loss = loss_simple / torch.exp(logvar_t) + logvar_t
        if self.learn_logvar:
            loss_dict[f'{prefix}/loss_gamma'] = loss.mean()
            loss_dict['logvar'] = self.logvar.data.mean()
        loss = self.l_simple_weight * loss.mean()
        loss_vlb = self.get_loss(model_output, target, mean=False).mean(dim=(1, 2, 3))
        loss_vlb = (self.lvlb_weights[t] * loss_vlb).mean()
        loss_dict[f'{prefix}/loss_vlb'] = loss_vlb
        loss = loss + self.original_elbo_weight * loss_vlb
        loss_dict[f'{prefix}/loss'] = loss
        return loss, loss_dict


class LatentFinetuneDiffusion(LatentDiffusion):
    """Keeps `concat_keys` / `finetune_keys` (reference 1512-1548); checkpoint surgery for widened input convs is a
    training-time concern and not reproduced."""

    def __init__(self, concat_keys: tuple, finetune_keys=("model.diffusion_model.input_blocks.0.0.weight",
                                                          "model_ema.diffusion_modelinput_blocks00weight"),
                 keep_finetune_dims=4, c_concat_log_start=None, c_concat_log_end=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.finetune_keys = finetune_keys
        self.concat_keys = concat_keys
        self.keep_dims = keep_finetune_dims
        self.c_concat_log_start = c_concat_log_start
        self.c_concat_log_end = c_concat_log_end


class LatentInpaintDiffusion(LatentFinetuneDiffusion):
    """Mask + masked-image concat conditioning, text via cross-attention (reference 1654-1701)."""

    def __init__(self, concat_keys=("mask", "masked_image"), masked_image_key="masked_image", *args, **kwargs):
        super().__init__(concat_keys, *args, **kwargs)
        self.masked_image_key = masked_image_key
        assert self.masked_image_key in concat_keys

    def get_input(self, batch, k, cond_key=None, bs=None, return_first_stage_outputs=False, force_c_encode=True):
        z, c, x, xrec, xc = super().get_input(batch, self.first_stage_key, return_first_stage_outputs=True,
                                              force_c_encode=force_c_encode, return_original_cond=True, bs=bs)
        assert exists(self.concat_keys)
        c_cat = []
        for ck in self.concat_keys:
            cc = batch[ck].permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()
            if bs is not None:
                cc = cc[:bs]
            cc = cc.to(self.device)
            if ck != self.masked_image_key:
                cc = torch.nn.functional.interpolate(cc, size=z.shape[-2:])      # mask: nearest to latent size
            else:
                cc = self.get_first_stage_encoding(self.encode_first_stage(cc))  # VAE(masked image) * scale_factor
            c_cat.append(cc)
        all_conds = {"c_concat": [torch.cat(c_cat, dim=1)], "c_crossattn": [c]}
        if return_first_stage_outputs:
            return z, all_conds, x, xrec, xc
        return z, all_conds
