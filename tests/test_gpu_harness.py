"""The evaluation harness end to end: tools/run_inpainting.py -- the CLI and call sequence of the reference's
test_inpainting.py -- EXECUTED as a subprocess against a synthetic `model_config.yaml`, a checkpoint under ckpts/epoch=*.ckpt
and a folder of pair directories read through the `dataloaders.test_dataset` drop-in; the prompt goes through the drop-in
PromptCLIPEmbedder on the open_clip stand-in (oracle/clip_stub.py), the latents through the HIP VAE / UNet / sampler."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, unet_ref, weights  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_config(path, size):
    import yaml
    cfg = G.CONFIGS["MID"]
    dd = dict(double_z=True, z_channels=4, resolution=size, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    model = {"target": "inpainting_ldm.ref_inpainting_ldm.RefInpaintLDM", "params": dict(
        linear_start=0.00085, linear_end=0.0120, timesteps=1000, first_stage_key="image", cond_stage_key="txt", channels=4,
        cond_stage_trainable=True, conditioning_key="hybrid", scale_factor=0.18215,
        data_config={"img_size": size, "repeat_sp_token": 4, "sp_token": "<special-token>"},
        unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": cfg.kwargs()},
        first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                            "params": {"ddconfig": dd, "embed_dim": 4, "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config={"target": "ldm.modules.encoders.Refill_modules.PromptCLIPEmbedder",
                           "params": dict(freeze=True, layer="penultimate", special_tokens=["repeat_4_<special-token>"],
                                          init_text=["reference on the left target on the right"])})}
    with open(path, "w") as f:
        yaml.safe_dump({"model": model}, f)


def test_run_inpainting_script_end_to_end(tmp_path):
    from PIL import Image
    size = 64
    mdir = tmp_path / "synthetic_model"
    (mdir / "ckpts").mkdir(parents=True)
    _write_config(str(mdir / "model_config.yaml"), size)
    # the open_clip stand-in must be importable inside the subprocess
    stub = tmp_path / "stubs"
    stub.mkdir()
    (stub / "open_clip.py").write_text("from oracle.clip_stub import *  # noqa: F401,F403  (test stand-in for the absent package)\n")
    # checkpoint: deterministic fills under the key names a Lightning checkpoint of the model has
    import leftrefill_amd.dropin as dropin
    dropin.install()
    sys.path.insert(0, str(stub))
    try:
        from inpainting_ldm.model import create_model
        model = create_model(str(mdir / "model_config.yaml"))
    finally:
        sys.path.remove(str(stub))
    sd = dict(model.state_dict())            # schedule buffers and the prompt encoder's (stand-in) weights as constructed
    for k, v in model.state_dict().items():
        if k.startswith("first_stage_model."):
            sd[k] = torch.from_numpy(weights.fill_like("vae2." + k[len("first_stage_model."):], v.shape)).to(v.dtype)
    for k, v in G.unet_state("MID").items():
        sd["model.diffusion_model." + k] = v
    torch.save({"state_dict": sd}, str(mdir / "ckpts" / "epoch=3.ckpt"))
    # data: three pair directories
    rng = np.random.RandomState(5)
    data = tmp_path / "pairs"
    for i in range(3):
        d = data / f"scene_{i}"
        d.mkdir(parents=True)
        for stem in ("source", "target"):
            Image.fromarray(rng.randint(0, 256, (96, 96, 3), dtype=np.uint8)).save(str(d / (stem + ".png")))
        m = np.zeros((96, 96), np.uint8)
        m[20:70, 30:80] = 255
        Image.fromarray(np.stack([m] * 3, -1)).save(str(d / "mask.png"))
    out_dir, met_dir = tmp_path / "out", tmp_path / "metrics"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(stub), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_inpainting.py"), "--model_path", str(mdir), "--test_path",
                        str(data), "--test_size", str(size), "--metric_size", str(size), "--batch_size", "2", "--cfg", "2.5", "--eta",
                        "1.0", "--output_path", str(out_dir), "--metric_output", str(met_dir)],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "WARNING" not in r.stdout, r.stdout[-1500:]
    lines = {ln.split(":")[0]: ln for ln in r.stdout.splitlines() if ln.startswith(("PSNR:", "SSIM:"))}
    psnr = float(lines["PSNR"].split()[1])
    ssim = float(lines["SSIM"].split()[1])
    print(r.stdout[-400:])
    assert "over 3 images" in lines["PSNR"] and np.isfinite(psnr) and 3.0 < psnr < 60.0 and -1.0 <= ssim <= 1.0
    pngs = sorted(os.listdir(str(out_dir)))
    assert len(pngs) == 3
    im = np.asarray(Image.open(str(out_dir / pngs[0])))
    assert im.shape == (size, size, 3)          # the right (target) half of the stitched canvas
    assert "PSNR" in (met_dir / "synthetic_model.txt").read_text()


def _write_mv_config(path, size, V, concat):
    import yaml
    cfg = G.mv_config(V, concat)
    canv = V - 1 if concat else V
    dd = dict(double_z=True, z_channels=4, resolution=size, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    model = {"target": "inpainting_ldm.multiview_ref_inpainting_ldm.RefInpaintLDM", "params": dict(
        linear_start=0.00085, linear_end=0.0120, timesteps=1000, first_stage_key="image", cond_stage_key="txt", channels=4,
        cond_stage_trainable=True, conditioning_key="hybrid", scale_factor=0.18215, view_mode=True, view_num=V, concat_target=concat,
        data_config={"img_size": size, "repeat_sp_token": 4, "sp_token": "<special-token>", "view_num": V, "view_token_len": 2,
                     "concat_target": concat},
        unet_config={"target": "ldm.modules.diffusionmodules.multiview_unet.MultiViewUnetModel", "params": cfg.kwargs()},
        first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                            "params": {"ddconfig": dd, "embed_dim": 4, "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config={"target": "ldm.modules.encoders.multiview_Refill_modules.PromptCLIPEmbedder",
                           "params": dict(freeze=True, layer="penultimate", special_tokens=["repeat_4_<special-token>"],
                                          init_text=["one of the photos of the same scene"], view_prompt=True, view_num=canv,
                                          view_token_len=2)})}
    with open(path, "w") as f:
        yaml.safe_dump({"model": model}, f)
    return cfg


@pytest.mark.parametrize("V,concat", [(3, True), (2, False)], ids=["v3_concat_target", "v2_plain"])
def test_run_inpainting_script_multiview(tmp_path, V, concat):
    """`tools/run_inpainting.py --multiview`: the call sequence of the reference's test_multiview_inpainting.py (77-233) EXECUTED as a
    subprocess on a synthetic multi-view model (MultiViewUnetModel + the per-view prompt encoder on the open_clip stand-in + HIP VAE):
    5-D batches, joint sampling of the (b v) canvases, the target view pasted with the mask of canvas 0, metrics and PNGs written.
    A last, smaller batch is split by the first batch's view count (reference 146-148)."""
    from PIL import Image
    size = 64
    mdir = tmp_path / "synthetic_mv_model"
    (mdir / "ckpts").mkdir(parents=True)
    cfg = _write_mv_config(str(mdir / "model_config.yaml"), size, V, concat)
    stub = tmp_path / "stubs"
    stub.mkdir()
    (stub / "open_clip.py").write_text("from oracle.clip_stub import *  # noqa: F401,F403  (test stand-in for the absent package)\n")
    import leftrefill_amd.dropin as dropin
    dropin.install()
    sys.path.insert(0, str(stub))
    try:
        from inpainting_ldm.model import create_model
        model = create_model(str(mdir / "model_config.yaml"))
    finally:
        sys.path.remove(str(stub))
    sd = dict(model.state_dict())
    for k, v in model.state_dict().items():
        if k.startswith("first_stage_model."):
            sd[k] = torch.from_numpy(weights.fill_like("vae2." + k[len("first_stage_model."):], v.shape)).to(v.dtype)
    for k, v in weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix="unet.MV.").items():
        sd["model.diffusion_model." + k] = v
    torch.save({"state_dict": sd}, str(mdir / "ckpts" / "epoch=1.ckpt"))
    out_dir, met_dir = tmp_path / "out", tmp_path / "metrics"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(stub), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_inpainting.py"), "--multiview", "--model_path", str(mdir),
                        "--synthetic", "2", "--test_size", str(size), "--metric_size", "32", "--batch_size", "2", "--cfg", "2.5",
                        "--eta", "1.0", "--output_path", str(out_dir), "--metric_output", str(met_dir)],
                       capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "WARNING" not in r.stdout, r.stdout[-1500:]
    lines = {ln.split(":")[0]: ln for ln in r.stdout.splitlines() if ln.startswith(("PSNR:", "SSIM:"))}
    psnr = float(lines["PSNR"].split()[1])
    ssim = float(lines["SSIM"].split()[1])
    assert "over 4 images" in lines["PSNR"] and np.isfinite(psnr) and 3.0 < psnr < 60.0 and -1.0 <= ssim <= 1.0
    pngs = sorted(os.listdir(str(out_dir)))
    assert len(pngs) == 4                        # one target view per sample, 2 batches x 2 samples
    assert np.asarray(Image.open(str(out_dir / pngs[0]))).shape == (32, 32, 3)      # target view, area-downsampled to --metric_size
    assert "PSNR" in (met_dir / "synthetic_mv_model.txt").read_text()
    # the dataset-backed form is refused, not silently replaced
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_inpainting.py"), "--multiview", "--model_path", str(mdir),
                         "--test_path", str(tmp_path)], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r2.returncode != 0 and "synthetic" in (r2.stdout + r2.stderr)
