#!/usr/bin/env python
"""Evaluation harness with the CLI and call sequence of the reference's test_inpainting.py (56-205), against the drop-in API.

    python tools/run_inpainting.py --model_path check_points/ref_guided_inpainting --cfg 2.5 --eta 1.0 [--test_path DIR]

Call sequence reproduced (reference line numbers): create_model(model_config.yaml).cpu() (82) -> load newest
ckpts/epoch=*.ckpt, then pretrained_models/512-inpainting-ema.ckpt when `save_prompt_only` (84-97) -> .to("cuda").eval()
(102-103) -> no_grad + autocast (126) -> model.log_images(batch, N, unconditional_guidance_scale=cfg, ddim_eta=eta) (141)
-> pred*mask + origin*(1-mask) (146-147) -> keep the right half (148-150) -> PSNR on (x+1)/2 (158), SSIM on the luma (160-162)
-> PNG (168-190).  LPIPS (159): `--lpips_weights alexnet-owt.pth,lpips_alex.pth` (torchvision's AlexNet + the lpips package's
v0.1 linear layers, or one state dict of `lpips.LPIPS(net='alex')`) feeds `evalglue.LPIPSAlex`; without the files (none ship:
no network) the metric is reported as not computed.

--test_path DIR is read through `dataloaders.test_dataset.TestInpaintingDataset` (drop-in of the reference loader: pair
directories with source / target / mask files) and torch's DataLoader, exactly like the reference (118-120).  Without it
(no dataset ships with the reference) `--synthetic N` builds N batches with the same batch contract
(dataloaders/test_dataset.py:91-105): image [B,512,1024,3] in [-1,1] (left reference | right target), mask [B,512,1024,1]
(left half 0), masked_image = image*(mask<0.5), txt = "<special-token0> ... <special-token49>".

--multiview: the call sequence of the reference's test_multiview_inpainting.py (77-233) for the multi-view task model
(inpainting_ldm.multiview_ref_inpainting_ldm.RefInpaintLDM over MultiViewUnetModel): 5-D batches [B, v, H, W, 3], log_images samples all
(b v) canvases jointly and returns the target view; the mask of canvas 0 of every sample pastes the known pixels back, a
[reference | target] canvas keeps its target half (`evalglue.compose_prediction_multiview`, reference 141-170).  Its dataset
(dataloaders/inpainting_crossview_dataset.py) is outside this build, so the mode runs on `--synthetic N` batches with that
batch contract: one prompt list per view, txt[view][batch].
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


def synthetic_mv_batches(n, batch_size, size, views, concat, view_token_len, sp_token="<special-token>", repeat=50, seed=0):
    """Multi-view batches: `views` canvases per sample; concat_target: canvas i = [reference_i | target] (size x 2 size), the mask marks a
    block pattern inside the target half; otherwise square views, view 0 is the (masked) target."""
    g = torch.Generator().manual_seed(seed)
    base = " ".join(f"{sp_token[:-1]}{i}>" for i in range(repeat))
    txt = [[base + " " + " ".join(f"<view_direct-{j}-{l}" for l in range(view_token_len))] * batch_size for j in range(views)]
    wc = 2 * size if concat else size
    for _ in range(n):
        img = torch.rand(batch_size, views, size, wc, 3, generator=g) * 2 - 1
        if concat:
            img[:, :, :, size:] = img[:, :1, :, size:]        # every canvas carries the same target on its right half
        mask = torch.zeros(batch_size, views, size, wc, 1)
        blocks = (torch.rand(batch_size, size // 32, size // 32, 1, generator=g) < 0.5).float()
        blocks[:, 0, 0] = 1.0                                 # never an empty mask (a tiny --test_size has only a few blocks)
        blocks = blocks.repeat_interleave(32, 1).repeat_interleave(32, 2)
        if concat:
            mask[:, :, :, size:, :] = blocks[:, None]
        else:
            mask[:, 0] = blocks
        yield {"image": img, "mask": mask, "masked_image": img * (mask < 0.5), "txt": [list(t) for t in txt]}


def dataset_batches(path, batch_size, size, model):
    """The reference's loader + DataLoader (test_inpainting.py:118-120)."""
    from torch.utils.data import DataLoader
    from dataloaders.test_dataset import TestInpaintingDataset
    cond_cfg = getattr(model, "cond_cfg", None) or {}
    data_cfg = dict(getattr(model, "data_cfg", None) or {})
    data_cfg.pop("img_size", None)
    ds = TestInpaintingDataset(path, img_size=size, deep_prompt=cond_cfg.get("deep_prompt", False), **data_cfg)
    return DataLoader(ds, batch_size=batch_size, shuffle=False)


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
    ap.add_argument("--metric_output", type=str, default="metric_outputs")
    ap.add_argument("--ngpu", type=int, default=1)       # parsed but unused, like the reference (62, 99)
    ap.add_argument("--fp16", action="store_true")       # idem (63)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--pretrained", type=str, default="pretrained_models/512-inpainting-ema.ckpt")
    ap.add_argument("--lpips_weights", type=str, default=None, help="comma-separated state-dict files for LPIPS(alex)")
    ap.add_argument("--multiview", action="store_true", help="multi-view task model: the call sequence of test_multiview_inpainting.py")
    a = ap.parse_args()
    if a.multiview and a.test_path:
        raise SystemExit("--multiview reads --synthetic batches only: the cross-view dataset loader is outside this build (SURVEY 2a)")

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
    if a.multiview:
        concat = bool(getattr(model, "concat_target", False))
        views = model.view_num - 1 if concat else model.view_num
        vlen = int((getattr(model, "cond_cfg", None) or {}).get("view_token_len", 0)) if (getattr(model, "cond_cfg", None) or {}).get("view_prompt", True) else 0
        dc = getattr(model, "data_cfg", None) or {}
        batches = synthetic_mv_batches(max(1, a.synthetic), a.batch_size, a.test_size, views, concat, vlen,
                                       sp_token=dc.get("sp_token", "<special-token>"), repeat=int(dc.get("repeat_sp_token", 50)))
    else:
        batches = dataset_batches(a.test_path, a.batch_size, a.test_size, model) if a.test_path else \
            synthetic_batches(max(1, a.synthetic), a.batch_size, a.test_size)
    global_view_num = 0
    lpips_fn = None
    if a.lpips_weights:
        lpips_fn = evalglue.LPIPSAlex().load_weights(*[torch.load(f, map_location="cpu") for f in a.lpips_weights.split(",")]).cuda()
    psnrs, ssims, lpipss = [], [], []
    with torch.no_grad(), torch.autocast("cuda"):
        for bi, batch in enumerate(batches):
            batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
            out = model.log_images(batch, batch["image"].shape[0], unconditional_guidance_scale=a.cfg, ddim_eta=a.eta)
            if not torch.isfinite(out["pred"]).all():
                bad = (~torch.isfinite(out["pred"])).float().mean().item()
                print(f"WARNING: {100 * bad:.2f} % of the decoded prediction is not finite (batch {bi})")
            if a.multiview:      # batch["mask"] was flattened to (b v) canvases by log_images, like the reference (100-105)
                pred, origin, global_view_num = evalglue.compose_prediction_multiview(out, batch["mask"], a.batch_size, global_view_num,
                                                                                      a.test_size, a.metric_size)
            else:
                pred, origin = evalglue.compose_prediction(out, batch["mask"], a.test_size, a.metric_size)
            psnrs.extend(evalglue.psnr01(pred, origin).tolist())
            if lpips_fn is not None:
                with torch.autocast("cuda", enabled=False):
                    lpipss.extend(lpips_fn(pred.float(), origin.float()).flatten().tolist())   # LPIPS takes [-1, 1] (159)
            for j in range(pred.shape[0]):
                ssims.append(evalglue.ssim_gray(evalglue.rgb_to_gray01(pred[j]), evalglue.rgb_to_gray01(origin[j])))
            p01 = (pred.float().clamp(-1, 1) + 1) / 2
            try:
                from PIL import Image
                for j in range(p01.shape[0]):
                    arr = (p01[j].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
                    Image.fromarray(arr).save(os.path.join(a.output_path, f"{bi:04d}_{j}.png"))
            except ImportError:
                pass
    print(f"PSNR: {float(np.mean(psnrs)):.3f} over {len(psnrs)} images")
    print(f"SSIM: {float(np.mean(ssims)):.4f}")
    lp = f"{float(np.mean(lpipss)):.4f}" if lpipss else "not computed (no --lpips_weights)"
    print(f"LPIPS: {lp}")
    os.makedirs(a.metric_output, exist_ok=True)
    with open(os.path.join(a.metric_output, os.path.basename(os.path.normpath(a.model_path)) + ".txt"), "w") as f:
        f.write(f"PSNR: {float(np.mean(psnrs)):.4f}\nSSIM: {float(np.mean(ssims)):.4f}\nLPIPS: {lp}\n")


if __name__ == "__main__":
    main()
