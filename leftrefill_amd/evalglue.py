"""Host-side glue of the evaluation harness (reference test_inpainting.py:143-166): paste the prediction into the known
pixels, keep the right (target) half of a stitched canvas, optional area down-sampling, PSNR on [0, 1].

Pure torch on whatever device the tensors live on.  SSIM restates `skimage.metrics.structural_similarity` of the pinned
scikit_image==0.18.1 with the defaults the reference call uses (test_inpainting.py:160-162); LPIPS needs the pretrained AlexNet
of the `lpips` package (no weights without network) and is not reproduced."""
import numpy as np
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


def rgb_to_gray01(x):
    """[3, H, W] in [-1, 1] -> [H, W] luma of (x + 1) / 2 (torchvision.transforms.functional.rgb_to_grayscale weights)."""
    p = (x.float() + 1) / 2
    return 0.2989 * p[0] + 0.587 * p[1] + 0.114 * p[2]


def ssim_gray(pred_gray, origin_gray):
    """Mean structural similarity of two float [H, W] images as `skimage.metrics.structural_similarity(a, b)` (0.18.1 defaults):
    7 x 7 uniform window, sample covariance (N / (N - 1)), K1 = 0.01, K2 = 0.03, data_range = 2 (what 0.18 takes for float
    inputs when none is given), float64 math, borders of (win - 1) / 2 pixels excluded from the mean."""
    from scipy.ndimage import uniform_filter
    a = np.asarray(pred_gray.detach().cpu() if torch.is_tensor(pred_gray) else pred_gray, dtype=np.float64)
    b = np.asarray(origin_gray.detach().cpu() if torch.is_tensor(origin_gray) else origin_gray, dtype=np.float64)
    win, k1, k2, rng = 7, 0.01, 0.03, 2.0
    n = win * win
    cov_norm = n / (n - 1.0)
    ua, ub = uniform_filter(a, size=win), uniform_filter(b, size=win)
    va = cov_norm * (uniform_filter(a * a, size=win) - ua * ua)
    vb = cov_norm * (uniform_filter(b * b, size=win) - ub * ub)
    vab = cov_norm * (uniform_filter(a * b, size=win) - ua * ub)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    s = ((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua * ua + ub * ub + c1) * (va + vb + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())
