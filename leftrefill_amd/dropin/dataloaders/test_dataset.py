"""`dataloaders.test_dataset.TestInpaintingDataset` (reference dataloaders/test_dataset.py:9-105), the loader
test_inpainting.py:118-120 builds: one sample per pair directory <root>/<pair>/{source,target}.{jpg,png} + mask.png (or a
text file listing pair directories; or masks cycled from `mask_path`).

Batch contract (91-105): image [H, 2H, 3] float32 in [-1, 1] = [source | target] resized to img_size, mask [H, 2H, 1] in
{0, 1} with the left (reference) half all zero, masked_image = image * (mask < 0.5), txt = the task prompt.

The reference decodes and resizes with OpenCV (un-vendored, absent here); this loader uses PIL for decoding and restates the
two resize modes in numpy: INTER_AREA as exact pixel-area averaging (what OpenCV computes when shrinking; the evaluation sets
are >= the 512-pixel test size) and INTER_NEAREST with OpenCV's source index floor(dst * scale).
"""
import os
from glob import glob

import numpy as np
from torch.utils.data import Dataset


def _area_weights(n_src, n_dst):
    """[n_dst, n_src] row-stochastic matrix: destination pixel i averages the source interval [i*r, (i+1)*r), r = n_src/n_dst."""
    r = n_src / n_dst
    w = np.zeros((n_dst, n_src), dtype=np.float64)
    for i in range(n_dst):
        lo, hi = i * r, (i + 1) * r
        for j in range(int(np.floor(lo)), min(n_src, int(np.ceil(hi)))):
            w[i, j] = max(0.0, min(hi, j + 1) - max(lo, j))
    return w / w.sum(axis=1, keepdims=True)


def resize_area(img, size):
    """uint8 [h, w, c] -> uint8 [size, size, c] by area averaging (cv2.resize(..., interpolation=cv2.INTER_AREA) when shrinking)."""
    h, w = img.shape[:2]
    if (h, w) == (size, size):
        return img
    out = np.einsum("ih,hwc->iwc", _area_weights(h, size), img.astype(np.float64))
    out = np.einsum("jw,iwc->ijc", _area_weights(w, size), out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_nearest(img, size):
    """[h, w] -> [size, size], source index floor(dst * src / dst_size) (cv2.INTER_NEAREST)."""
    h, w = img.shape[:2]
    yi = np.minimum((np.arange(size) * (h / size)).astype(np.int64), h - 1)
    xi = np.minimum((np.arange(size) * (w / size)).astype(np.int64), w - 1)
    return img[yi][:, xi]


def _read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


class TestInpaintingDataset(Dataset):
    __test__ = False      # not a pytest class

    def __init__(self, root_path, img_size=256, token_map=None, mask_path=None, **kwargs):
        self.img_size = img_size
        self.root_path = root_path
        self.token_map = token_map
        if os.path.isdir(root_path):
            self.pairs = sorted(glob(root_path + "/*"), key=lambda p: p.split("/")[-1])
        else:
            with open(root_path) as f:
                self.pairs = [ln.strip() for ln in f.readlines()]
        self.mask_list = None
        if mask_path is not None:
            self.mask_list = sorted(glob(mask_path + "/*"), key=lambda p: p.split("/")[-1])
        self.repeat_sp_token = kwargs.get("repeat_sp_token", 0)   # > 0: the prompt is the special token repeated that often
        self.sp_token = kwargs.get("sp_token", None)
        self.deep_prompt = kwargs.get("deep_prompt", False)
        self.cross_attn_layers = 16

    def __len__(self):
        return len(self.pairs)

    def resize_and_crop(self, image):
        return resize_area(image, self.img_size)

    def get_prompt(self):
        if self.repeat_sp_token > 0 and self.sp_token is not None:
            text = " ".join(self.sp_token.replace(">", f"{i}>") for i in range(self.repeat_sp_token))
            if self.deep_prompt:        # one prompt per cross-attention layer
                return [text.replace(">", f"-layer{layer}>") for layer in range(self.cross_attn_layers)]
            return text
        if self.token_map is None:
            return "[REFERENCE_INPAINTING]"
        t = self.token_map
        return f"Both {t['left_token']} and {t['right_token']} images show the {t['real_token']} with different {t['task_token']}."

    def __getitem__(self, idx):
        pair = self.pairs[idx]

        def pick(stem):
            p = f"{pair}/{stem}.jpg"
            return p if os.path.exists(p) else p.replace(".jpg", ".png")

        source = self.resize_and_crop(_read_rgb(pick("source")))
        target = self.resize_and_crop(_read_rgb(pick("target")))
        image = np.concatenate([source, target], axis=1).astype(np.float32) / 127.5 - 1.0
        mask_file = f"{pair}/mask.png" if self.mask_list is None else self.mask_list[idx % len(self.mask_list)]
        mask = _read_rgb(mask_file)[:, :, 2]       # cv2.imread(...)[:, :, 0] is the BLUE plane of the file
        mask = resize_nearest(mask, self.img_size).astype(np.float32) / 255.0
        mask = np.concatenate([np.zeros_like(mask), mask], axis=1)[:, :, None]
        return dict(image=image, txt=self.get_prompt(), masked_image=image * (mask < 0.5), mask=mask)
