"""`inpainting_ldm.multiview_ref_inpainting_ldm.RefInpaintLDM` -- multi-view task model (configs/multiview_ref_inpainting.yaml:2).

Inference entry points of the reference class (inpainting_ldm/multiview_ref_inpainting_ldm.py): the extra constructor
kwargs `view_mode / view_num / concat_target / reduced_loss` (33-36), `get_input` flattening `[b, v, h, w, c]` batches
to `(b v)` canvases (99-111) and `log_images` (113-180), which samples all `(b v)` canvases jointly (the views meet in
MultiViewUnetModel's re-arranged self-attention) and returns the target view plus the references:
  concat_target: canvases are [ref_i | target]; pred / origin / masked = right half of canvas 0, reference = left halves;
  otherwise    : view 0 is the target, views 1.. are the references.
Training (`p_losses`, 38-91) is out of scope.
"""
import torch

from inpainting_ldm.ref_inpainting_ldm import RefInpaintLDM as _SingleViewLDM


class RefInpaintLDM(_SingleViewLDM):
    def __init__(self, *args, **kwargs):
        mv = (kwargs.get('view_mode', False), kwargs.get('view_num', 4), kwargs.get('concat_target', False),
              kwargs.get('reduced_loss', True))
        super().__init__(*args, **kwargs)       # DDPM.__init__ swallows the extra keys, as in the reference
        self.view_mode, self.view_num, self.concat_target, self.reduced_loss = mv

    def p_losses(self, *args, **kwargs):
        raise NotImplementedError("training is outside the MI355X sampling path")

    def get_input(self, batch, k, cond_key=None, bs=None, return_first_stage_outputs=False, force_c_encode=True):
        if batch['image'].dim() == 5:            # [b, v, h, w, c] -> (b v) canvases, in place like the reference (100-104)
            for key in ('image', 'masked_image', 'mask'):
                t = batch[key]
                batch[key] = t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])
        return super().get_input(batch, k, cond_key=cond_key, bs=bs,
                                 return_first_stage_outputs=return_first_stage_outputs, force_c_encode=force_c_encode)

    @torch.no_grad()
    def log_images(self, batch, N=4, ddim_steps=50, ddim_eta=0.0, unconditional_guidance_scale=9.0, **kwargs):
        img = batch['image']
        N = img.shape[0] * img.shape[1] if img.dim() == 5 else img.shape[0]      # every canvas of every sample (114-117)
        v = self.view_num - 1 if self.concat_target else self.view_num
        use_ddim = ddim_steps is not None
        z, c = self.get_input(batch, self.first_stage_key, bs=N)
        c_concat, c_crossattn = c["c_concat"][0][:N], c["c_crossattn"][0][:N]
        N = min(z.shape[0], N)
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
        pred = self.decode_first_stage(samples)

        def by_view(t):      # '(b v) c h w -> b v c h w'
            return t.reshape(t.shape[0] // v, v, *t.shape[1:])

        masked = by_view(batch['masked_image'].permute(0, 3, 1, 2))
        origin = by_view(batch['image'].permute(0, 3, 1, 2))
        pred = by_view(pred)
        log = dict()
        if self.concat_target:
            cut = pred.shape[3]          # the reference splits the width at pred.shape[3] (= H: square halves), 163-166
            log["reference"] = masked[:, :, :, :, 0:cut]
            log["masked_image"] = masked[:, 0, :, :, cut:]
            log["origin_image"] = origin[:, 0, :, :, cut:]
            log["pred"] = pred[:, 0, :, :, cut:]
        else:
            log["reference"] = masked[:, 1:]
            log["masked_image"] = masked[:, 0]
            log["origin_image"] = origin[:, 0]
            log["pred"] = pred[:, 0]
        return log
