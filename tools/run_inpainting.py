#!/usr/bin/env python
"""Evaluation harness with the CLI and call sequence of the reference's test_inpainting.py (56-205), against the drop-in API.

    python tools/run_inpainting.py --model_path check_points/ref_guided_inpainting --cfg 2.5 --eta 1.0 [--test_path DIR]

Call sequence reproduced (reference line numbers): create_model(model_config.yaml).cpu() (82) -> load newest
ckpts/epoch=*.ckpt, then pretrained_models/512-inpainting-ema.ckpt when `save_prompt_only` (84-97) -> .to("cuda").eval()
(102-103) -> no_grad + autocast (126) -> model.log_images(batch, N, unconditional_guidance_scale=cfg, ddim_eta=eta) (141)
-> pred*mask + origin*(1-mask) (146-147) -> keep the right half (148-150) -> PSNR on (x+1)/2 (158) -> PNG (168-190).
LPIPS / SSIM need third-party packages that are out of scope; PSNR is computed directly.

Without --test_path (no dataset ships with the reference) `--synthetic N` builds N batches with the batch contract of
dataloaders/test_dataset.py:91-105: image [B,512,1024,3] in [-1,1] (left reference | right target), mask [B,512,1024,1]
(left half 0), masked_image = image*(mask<0.5), txt = "<special-token0> ... <special-token49>".
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_batches(n, batch_size, size, sp_token="<special-token>", repeat=50, seed=0):
    g = torch.Generator().manual_seed(seed)
    txt = " ".join(f"{sp_token[:-1]}{i}>" for i in range(repeat))
    for _ in range(n):
        img = torch.rand(batch_size, size, 2 * size, 3, generator=g) * 2 - 1
        mask = torch.zeros(batch_size, size, 2 * size, 1)
        blocks = (torch.rand(batch_size, size // 32, size // 32, 1, generator=g) < 0.5).float()
        mask[:, :, size:, :] = blocks.repeat_interleave(32, 1).repeat_interleave(32, 2)
        yield {"image": img, "mask": mask, "masked_image": img * (mask < 0.5), "txt": [txt] * batch_size}


def folder_batches(path, batch_size, size):
    """<path>/*_ref.png + *_tgt.png + *_mask.png triplets, loaded with PIL (the reference uses cv2, test_dataset.py:62-90)."""
    from PIL import Image
    refs = sorted(glob.glob(os.path.join(path, "*_ref.png")))
    items = []
    for r in refs:
        t, m = r.replace("_ref.png", "_tgt.png"), r.replace("_ref.png", "_mask.png")
        ld = lambda p, mode: np.asarray(Image.open(p).convert(mode).resize((size, size)), dtype=np.float32) / 255.0
        img = np.concatenate([ld(r, "RGB"), ld(t, "RGB")], axis=1) * 2 - 1
        mask = np.concatenate([np.zeros((size, size, 1), np.float32), (ld(m, "L")[..., None] > 0.5).astype(np.float32)], 1)
        items.append((img, mask))
    for i in range(0, len(items), batch_size):
        img = torch.from_numpy(np.stack([a for a, _ in items[i:i + batch_size]]))
        mask = torch.from_numpy(np.stack([b for _, b in items[i:i + batch_size]]))
        txt = " ".join(f"<special-token{j}>" for j in range(50))
        yield {"image": img, "mask": mask, "masked_image": img * (mask < 0.5), "txt": [txt] * img.shape[0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", type=str, required=True)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--test_path", type=str, default=None)
    ap.add_argument("--cfg", type=float, default=2.5)
    ap.add_argument("--test_size", type=int, default=512)
    ap.add_argument("--metric_size", type=int, default=512)
    ap.add_argument("--eta", type=float, default=1.0)
    ap.add_argument("--output_path", type=str, default="outputs")
    ap.add_argument("--ngpu", type=int, default=1)       # parsed but unused, like the reference (62, 99)
    ap.add_argument("--fp16", action="store_true")       # idem (63)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--pretrained", type=str, default="pretrained_models/512-inpainting-ema.ckpt")
    a = ap.parse_args()

    import leftrefill_amd.dropin as dropin
    dropin.install()
    from inpainting_ldm.model import create_model, load_state_dict
    from leftrefill_amd import evalglue

    model = create_model(os.path.join(a.model_path, "model_config.yaml")).cpu()
    ckpts = sorted(glob.glob(os.path.join(a.model_path, "ckpts", "epoch=*.ckpt")), key=os.path.getmtime)
    if ckpts:
        print(model.load_state_dict(load_state_dict(ckpts[-1]), strict=False))
    if getattr(model, "save_prompt_only", False) and os.path.exists(a.pretrained):
        print(model.load_state_dict(load_state_dict(a.pretrained), strict=False))
    model = model.to("cuda").eval()
    os.makedirs(a.output_path, exist_ok=True)
    batches = folder_batches(a.test_path, a.batch_size, a.test_size) if a.test_path else \
        synthetic_batches(max(1, a.synthetic), a.batch_size, a.test_size)
    psnrs = []
    with torch.no_grad(), torch.autocast("cuda"):
        for bi, batch in enumerate(batches):
            batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
            out = model.log_images(batch, batch["image"].shape[0], unconditional_guidance_scale=a.cfg, ddim_eta=a.eta)
            pred, origin = evalglue.compose_prediction(out, batch["mask"], a.test_size, a.metric_size)
            psnrs.extend(evalglue.psnr01(pred, origin).tolist())
            p01 = (pred.float().clamp(-1, 1) + 1) / 2
            try:
                from PIL import Image
                for j in range(p01.shape[0]):
                    arr = (p01[j].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
                    Image.fromarray(arr).save(os.path.join(a.output_path, f"{bi:04d}_{j}.png"))
            except ImportError:
                pass
    print(f"PSNR: {float(np.mean(psnrs)):.3f} over {len(psnrs)} images")
    os.makedirs("metric_outputs", exist_ok=True)
    with open(os.path.join("metric_outputs", os.path.basename(os.path.normpath(a.model_path)) + ".txt"), "w") as f:
        f.write(f"PSNR: {float(np.mean(psnrs)):.4f}\n")


if __name__ == "__main__":
    main()
