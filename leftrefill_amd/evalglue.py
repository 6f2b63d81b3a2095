"""Host-side glue of the evaluation harness (reference test_inpainting.py:143-166): paste the prediction into the known
pixels, keep the right (target) half of a stitched canvas, optional area down-sampling, PSNR on [0, 1].

Pure torch on whatever device the tensors live on.  SSIM restates `skimage.metrics.structural_similarity` of the pinned
scikit_image==0.18.1 with the defaults the reference call uses (test_inpainting.py:160-162); `LPIPSAlex` restates the published
LPIPS(alex) computation (test_inpainting.py:159) and takes the pretrained weights from the user (none ship without network;
parity-unpinned: the `lpips` package is absent here)."""
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


def compose_prediction_multiview(out, mask_flat_nhwc, batch_size, global_view_num=0, test_size=None, metric_size=None):
    """The multi-view harness' composition (reference test_multiview_inpainting.py:141-170).  out: dict from
    multiview_ref_inpainting_ldm.RefInpaintLDM.log_images (pred / origin_image [b, 3, h, w]: the target view only);
    mask_flat_nhwc: batch['mask'] AFTER log_images flattened it in place to [(b v), H, W, 1] (reference get_input, 100-105);
    batch_size: the loader's batch size.  Returns (pred, origin, global_view_num):
      view_num = rows / batch_size, remembered from the FIRST batch (`global_view_num`, 146-148: a last, smaller batch is split by it);
      mask = the mask of canvas 0 of every sample (149-151); a non-square canvas keeps the columns from H on (152-153: the target half
      of a [reference | target] canvas);  pred = pred * mask + origin * (1 - mask) (155);  h != w -> keep columns w//2: (156-158);
      float32 (160-162);  metric_size < test_size -> F.interpolate(mode='area') (164-168)."""
    mask = mask_flat_nhwc.permute(0, 3, 1, 2)
    view_num = int(mask.shape[0] / batch_size)
    if global_view_num == 0:
        global_view_num = view_num
    real_bs = int(mask.shape[0] / global_view_num)
    mask = mask.reshape(real_bs, global_view_num, *mask.shape[1:])[:, 0]
    if mask.shape[3] != mask.shape[2]:
        mask = mask[:, :, :, mask.shape[2]:]
    h, w = out["pred"].shape[2], out["pred"].shape[3]
    # composited in fp32 like the reference (the float32 mask promotes the product, test_multiview_inpainting.py:159-162)
    mask = mask.float()
    pred = out["pred"].float() * mask + out["origin_image"].float() * (1 - mask)
    origin = out["origin_image"]
    if h != w:
        pred, origin = pred[:, :, :, w // 2:], origin[:, :, :, w // 2:]
    pred, origin = pred.float(), origin.float()
    if metric_size is not None and test_size is not None and metric_size < test_size:
        pred = F.interpolate(pred, size=(metric_size, metric_size), mode="area")
        origin = F.interpolate(origin, size=(metric_size, metric_size), mode="area")
    return pred, origin, global_view_num


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


class LPIPSAlex(torch.nn.Module):
    """Learned perceptual distance of the harness (`model.loss_fn_alex = lpips.LPIPS(net='alex')`, test_inpainting.py:159).

    The `lpips` package (requirements: lpips==0.1.4) is a third-party dependency that is absent from the reference tree and
    from this image, together with its weights; this class restates the PUBLISHED algorithm (Zhang et al., CVPR 2018, v0.1
    linear calibration) and is **parity-unpinned**: inputs in [-1, 1] -> fixed per-channel shift / scale -> the five ReLU stages of
    torchvision's AlexNet `features` -> each stage's activations normalised to unit length over channels (eps 1e-10 added to the
    norm) -> squared difference -> non-negative 1x1 `lin` weights -> spatial mean -> sum over the five stages.  Output [N, 1, 1, 1].

    No weights ship with this repo (no network); `load_weights` takes the two files a user of the reference already has --
    torchvision's `alexnet-owt-*.pth` (`features.{0,3,6,8,10}.{weight,bias}`) and lpips' `weights/v0.1/alex.pth`
    (`lin{0..4}.model.1.weight`) -- or one state dict saved from `lpips.LPIPS(net='alex')` (`net.slice{1..5}.{0,3,6,8,10}.*`,
    `lin{k}.model.1.weight`).  Without weights `forward` raises: the harness then reports LPIPS as not computed."""

    SHIFT = (-0.030, -0.088, -0.188)
    SCALE = (0.458, 0.448, 0.450)
    CONVS = ((3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1))   # cin, cout, k, stride, pad
    FEATURE_INDEX = (0, 3, 6, 8, 10)          # positions of the convolutions in torchvision's alexnet.features
    POOL_BEFORE = (False, True, True, False, False)   # MaxPool2d(3, 2) in front of conv2 and conv3

    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor(self.SHIFT).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor(self.SCALE).view(1, 3, 1, 1))
        self.convs = torch.nn.ModuleList(torch.nn.Conv2d(ci, co, k, s, p) for ci, co, k, s, p in self.CONVS)
        self.lins = torch.nn.ParameterList(torch.nn.Parameter(torch.zeros(1, co, 1, 1), requires_grad=False) for _, co, *_ in self.CONVS)
        self.loaded = False
        for p in self.parameters():
            p.requires_grad_(False)

    def load_weights(self, *state_dicts):
        """Accepts any mix of the key spellings in the class docstring; raises KeyError naming what is still missing."""
        sd = {}
        for s in state_dicts:
            sd.update(s)
        missing = []
        for i, fi in enumerate(self.FEATURE_INDEX):
            for leaf in ("weight", "bias"):
                for key in (f"features.{fi}.{leaf}", f"net.slice{i + 1}.{fi}.{leaf}"):
                    if key in sd:
                        getattr(self.convs[i], leaf).copy_(sd[key])
                        break
                else:
                    missing.append(f"features.{fi}.{leaf}")
            for key in (f"lin{i}.model.1.weight", f"lins.{i}.model.1.weight"):
                if key in sd:
                    self.lins[i].copy_(sd[key].reshape(1, -1, 1, 1))
                    break
            else:
                missing.append(f"lin{i}.model.1.weight")
        if missing:
            raise KeyError("LPIPS(alex) weights missing: " + ", ".join(missing))
        self.loaded = True
        return self

    def features(self, x):
        outs = []
        h = (x - self.shift) / self.scale
        for conv, pool in zip(self.convs, self.POOL_BEFORE):
            if pool:
                h = F.max_pool2d(h, 3, 2)
            h = F.relu(conv(h))
            outs.append(h)
        return outs

    @torch.no_grad()
    def forward(self, a, b):
        if not self.loaded:
            raise RuntimeError("LPIPSAlex has no weights: call load_weights(alexnet_state_dict, lpips_alex_state_dict) first")
        a, b = a.float(), b.float()
        total = 0
        for fa, fb, lin in zip(self.features(a), self.features(b), self.lins):
            na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + ((na - nb) ** 2 * lin).sum(1, keepdim=True).mean((2, 3), keepdim=True)
        return total
