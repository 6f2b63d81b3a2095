"""Host-side glue of the evaluation harness (reference test_inpainting.py:143-166): paste the prediction into the known
pixels, keep the right (target) half of a stitched canvas, optional area down-sampling, PSNR on [0, 1].

Pure torch on whatever device the tensors live on; LPIPS / SSIM come from third-party packages in the reference
(lpips, skimage) and are not reproduced."""
import torch
import torch.nn.functional as F


def compose_prediction(out, mask_nhwc, test_size=None, metric_size=None):
    """out: dict from RefInpaintLDM.log_images ({'pred', 'origin_image', 'masked_image'}, NCHW in [-1, 1]);
    mask_nhwc: batch['mask'] [N, H, W, 1].  Returns (pred, origin) as the reference evaluates them:
      pred = pred * mask + origin * (1 - mask)            (146)
      h != w -> keep columns w//2:                        (147-149)
      metric_size < test_size -> F.interpolate(mode='area') to (metric_size, metric_size)   (151-155)"""
    mask = mask_nhwc.permute(0, 3, 1, 2).to(out["pred"].dtype)
    pred = out["pred"] * mask + out["origin_image"].to(out["pred"].dtype) * (1 - mask)
    origin = out["origin_image"]
    h, w = pred.shape[2], pred.shape[3]
    if h != w:
        pred, origin = pred[:, :, :, w // 2:], origin[:, :, :, w // 2:]
    if metric_size is not None and test_size is not None and metric_size < test_size:
        pred = F.interpolate(pred, size=(metric_size, metric_size), mode="area")
        origin = F.interpolate(origin, size=(metric_size, metric_size), mode="area")
    return pred, origin


def psnr01(pred, origin):
    """Per-image PSNR of (x + 1) / 2 with data_range 1.0, as torchmetrics.functional.peak_signal_noise_ratio computes it
    for one image at a time (158): 10 log10(1 / mse); no clamping."""
    p, o = (pred.float() + 1) / 2, (origin.float() + 1) / 2
    mse = ((p - o) ** 2).flatten(1).mean(1)
    return 10.0 * torch.log10(1.0 / mse)
