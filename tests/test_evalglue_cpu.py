"""Caller glue either side of the hot path, CPU: the drop-in `dataloaders.test_dataset` loader and the
SSIM restatement, both checked against independent brute-force computations."""
import os

import numpy as np
import pytest
import torch


def _write_pairs(root, n=3, hw=(96, 80), seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    for i in range(n):
        d = os.path.join(root, f"pair_{i:02d}")
        os.makedirs(d)
        for stem in ("source", "target"):
            Image.fromarray(rng.randint(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)).save(os.path.join(d, stem + ".png"))
        m = (rng.rand(hw[0], hw[1]) < 0.4).astype(np.uint8) * 255
        Image.fromarray(np.stack([m, m, m], -1)).save(os.path.join(d, "mask.png"))
    return root


def test_dataset_contract_and_prompts(tmp_path):
    import leftrefill_amd.dropin as dropin
    dropin.install()
    from dataloaders.test_dataset import TestInpaintingDataset, resize_area, resize_nearest
    from PIL import Image
    root = _write_pairs(str(tmp_path / "pairs"))
    ds = TestInpaintingDataset(root, img_size=32, repeat_sp_token=3, sp_token="<special-token>")
    assert len(ds) == 3
    it = ds[1]
    assert it["image"].shape == (32, 64, 3) and it["image"].dtype == np.float32
    assert it["mask"].shape == (32, 64, 1) and set(np.unique(it["mask"])) <= {0.0, 1.0}
    assert it["image"].min() >= -1.0 and it["image"].max() <= 1.0
    assert (it["mask"][:, :32] == 0).all() and it["mask"][:, 32:].mean() > 0.1      # left = reference: never masked
    assert np.array_equal(it["masked_image"], it["image"] * (it["mask"] < 0.5))
    assert it["txt"] == "<special-token0> <special-token1> <special-token2>"
    # area resize == exact average over the source footprint (96 -> 32 rows: 3 rows each; 80 -> 32 columns: 2.5 columns)
    src = np.asarray(Image.open(os.path.join(root, "pair_01", "source.png")).convert("RGB")).astype(np.float64)
    rows = src.reshape(32, 3, 80, 3).mean(1)
    col0 = (rows[:, 0] + rows[:, 1] + 0.5 * rows[:, 2]) / 2.5
    got = (it["image"][:, 0] + 1.0) * 127.5
    assert np.abs(got - np.rint(col0)).max() <= 1.0 + 1e-3
    sq = src.astype(np.uint8)[:80, :80]
    assert np.array_equal(resize_area(sq, 80), sq)          # identity size: untouched
    # nearest: source index floor(dst * scale)
    m = np.arange(96 * 80).reshape(96, 80)
    r = resize_nearest(m, 32)
    assert r[5, 7] == m[15, int(7 * 2.5)] and r[31, 31] == m[93, 77]
    # batching through torch's DataLoader, as the harness does
    from torch.utils.data import DataLoader
    b = next(iter(DataLoader(ds, batch_size=2, shuffle=False)))
    assert b["image"].shape == (2, 32, 64, 3) and b["mask"].shape == (2, 32, 64, 1) and len(b["txt"]) == 2
    # deep prompts: one prompt per cross-attention layer; token-map prompt; legacy prompt; pair list file
    dp = TestInpaintingDataset(root, img_size=32, repeat_sp_token=2, sp_token="<special-token>", deep_prompt=True).get_prompt()
    assert len(dp) == 16 and dp[3] == "<special-token0-layer3> <special-token1-layer3>"
    tm = dict(left_token="<left>", right_token="<right>", task_token="<task>", real_token="<real>")
    assert TestInpaintingDataset(root, token_map=tm).get_prompt() == "Both <left> and <right> images show the <real> with different <task>."
    assert TestInpaintingDataset(root).get_prompt() == "[REFERENCE_INPAINTING]"
    lst = tmp_path / "pairs.txt"
    lst.write_text("\n".join(sorted(os.path.join(root, d) for d in os.listdir(root))) + "\n")
    assert len(TestInpaintingDataset(str(lst), img_size=32)) == 3


def test_ssim_matches_bruteforce_definition():
    from leftrefill_amd import evalglue
    rng = np.random.RandomState(1)
    a = rng.rand(24, 30).astype(np.float32)
    b = np.clip(a + 0.1 * rng.randn(24, 30), 0, 1).astype(np.float32)
    assert abs(evalglue.ssim_gray(torch.from_numpy(a), torch.from_numpy(a)) - 1.0) < 1e-12
    # brute force over every fully-inside 7x7 window (the border windows are exactly the ones the mean excludes)
    c1, c2 = (0.01 * 2.0) ** 2, (0.03 * 2.0) ** 2
    vals = []
    for y in range(3, 24 - 3):
        for x in range(3, 30 - 3):
            pa, pb = a[y - 3:y + 4, x - 3:x + 4].astype(np.float64), b[y - 3:y + 4, x - 3:x + 4].astype(np.float64)
            ua, ub = pa.mean(), pb.mean()
            va, vb = pa.var(ddof=1), pb.var(ddof=1)
            vab = ((pa - ua) * (pb - ub)).sum() / 48.0
            vals.append(((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua * ua + ub * ub + c1) * (va + vb + c2)))
    assert abs(evalglue.ssim_gray(torch.from_numpy(a), torch.from_numpy(b)) - np.mean(vals)) < 1e-9
    g = evalglue.rgb_to_gray01(torch.tensor([[[1.0]], [[-1.0]], [[0.0]]]))
    assert abs(g.item() - (0.2989 * 1.0 + 0.587 * 0.0 + 0.114 * 0.5)) < 1e-6


def test_dropin_dataloaders_does_not_shadow_other_dataset_modules(tmp_path, monkeypatch):
    """the reference's `dataloaders` directory also holds the training / multi-view datasets; after install()
    those must still import while `dataloaders.test_dataset` resolves to the drop-in."""
    import importlib
    import sys
    other = tmp_path / "refroot" / "dataloaders"
    other.mkdir(parents=True)
    (other / "inpainting_crossview_dataset.py").write_text("MARK = 'reference module'\n")
    (other / "test_dataset.py").write_text("MARK = 'reference test_dataset'\n")
    monkeypatch.syspath_prepend(str(tmp_path / "refroot"))
    for name in [n for n in sys.modules if n == "dataloaders" or n.startswith("dataloaders.")]:
        monkeypatch.delitem(sys.modules, name)
    from leftrefill_amd.dropin import install
    root = install()
    mod = importlib.import_module("dataloaders.inpainting_crossview_dataset")
    assert mod.MARK == "reference module"
    td = importlib.import_module("dataloaders.test_dataset")
    assert td.__file__.startswith(root) and hasattr(td, "TestInpaintingDataset")
    for name in [n for n in sys.modules if n == "dataloaders" or n.startswith("dataloaders.")]:
        monkeypatch.delitem(sys.modules, name)


def test_harness_composition_hand_derived_fixture():
    """VERDICT r2 missing #6: the paste / crop / area-downsample / PSNR composition of reference test_inpainting.py:146-158 against
    a HAND-DERIVED fixture (tests/golden/harness_fixture.json; the reference script itself cannot run here).  Derivation:
      canvas 4 x 8, pred = 0.5 everywhere, origin = -0.5 on the left half / 0.25 on the right half, mask = 1 in columns 6, 7;
      paste (146): pred * mask + origin * (1 - mask) -> columns 0-3: -0.5, 4-5: 0.25, 6-7: 0.5;
      h != w (147-149): keep columns 4.. -> rows [0.25, 0.25, 0.5, 0.5]; origin rows [0.25] * 4;
      metric_size 2 < test_size 4 (151-153): 'area' = mean of each 2 x 2 block -> pred rows [0.25, 0.5], origin 0.25;
      PSNR (158) on (x + 1) / 2: pred 0.625 / 0.75, origin 0.625 -> squared errors 0 and 0.125^2, mse = 0.0078125 = 2^-7,
      10 log10(2^7) = 21.0721 dB (the same without down-sampling: half of the pixels differ by 0.125 either way);
      luma (160): 0.2989 + 0.587 + 0.114 = 0.9999 times the grey level."""
    import json
    from leftrefill_amd import evalglue
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness_fixture.json")))
    h, w, c = fx["h"], fx["w"], fx["channels"]
    pred = torch.full((1, c, h, w), fx["pred_value"])
    origin = torch.cat([torch.full((1, c, h, w // 2), fx["origin_left_value"]), torch.full((1, c, h, w // 2), fx["origin_right_value"])], dim=3)
    mask = torch.zeros(1, h, w, 1)
    mask[:, :, fx["mask_columns"], :] = 1.0
    out = {"pred": pred, "origin_image": origin, "masked_image": origin * (1 - mask.permute(0, 3, 1, 2))}
    p_full, o_full = evalglue.compose_prediction(out, mask)
    assert p_full.shape == (1, c, h, w // 2) and o_full.shape == p_full.shape
    assert torch.allclose(p_full[0, 1, 2], torch.tensor(fx["expected_pred_after_paste_right_half_row"]))
    assert abs(evalglue.psnr01(p_full, o_full).item() - fx["expected_psnr_without_downsampling_db"]) < 1e-4
    p, o = evalglue.compose_prediction(out, mask, test_size=fx["test_size"], metric_size=fx["metric_size"])
    assert p.shape == (1, c, 2, 2)
    assert torch.allclose(p[0, 0, 1], torch.tensor(fx["expected_pred_metric_row"])) and torch.allclose(o, torch.full_like(o, fx["expected_origin_metric_value"]))
    mse = (((p + 1) / 2 - (o + 1) / 2) ** 2).mean().item()
    assert abs(mse - fx["expected_mse01"]) < 1e-9
    assert abs(evalglue.psnr01(p, o).item() - fx["expected_psnr_db"]) < 1e-4
    assert torch.allclose(evalglue.rgb_to_gray01(p[0])[0], torch.tensor(fx["expected_gray_pred_metric_row"]), atol=1e-6)


def test_multiview_harness_composition_hand_derived():
    """The multi-view harness' composition (reference test_multiview_inpainting.py:141-170) on hand-derived numbers; the reference
    script cannot run here (its dataset module and metric packages are absent), so this pins the RESTATEMENT against a derivation:
      loader batch size 2, 3 canvases per sample, canvases [reference | target] of 2 x 4 pixels (concat_target);
      batch['mask'] after log_images flattened it: [(b v) = 6, 2, 4, 1]; canvas 0 of sample s has mask = 1 at target column 1 (canvas
      column 3) for s = 0 and at target column 0 (canvas column 2) for s = 1; the masks of canvases 1, 2 are all ones and must be IGNORED
      (149-151 take canvas 0); non-square canvas -> columns 2.. (152-153);
      log_images returns the target halves: pred = 0.75 everywhere, origin = -0.25 -> pasted rows: sample 0 [-0.25, 0.75], sample 1
      [0.75, -0.25] (155); square already, so no second crop (156-158);
      a LAST batch with one sample (3 rows) is split by the view count of the FIRST batch (146-148: 3 / 3 = 1 sample), not by
      rows / batch_size = 1.5."""
    from leftrefill_amd import evalglue
    b, v, h = 2, 3, 2
    mask = torch.ones(b * v, h, 2 * h, 1)
    mask[0] = 0
    mask[0, :, 3] = 1
    mask[3] = 0
    mask[3, :, 2] = 1
    out = {"pred": torch.full((b, 3, h, h), 0.75), "origin_image": torch.full((b, 3, h, h), -0.25)}
    pred, origin, gv = evalglue.compose_prediction_multiview(out, mask, batch_size=2)
    assert gv == 3 and pred.shape == (b, 3, h, h) and pred.dtype == torch.float32
    assert torch.equal(pred[0, 1], torch.tensor([[-0.25, 0.75]] * h)) and torch.equal(pred[1, 2], torch.tensor([[0.75, -0.25]] * h))
    assert torch.equal(origin, out["origin_image"])
    # PSNR on (x + 1) / 2: half of the pixels differ by 0.5 -> mse = 0.125 -> 10 log10(8) = 9.0309 dB
    assert abs(evalglue.psnr01(pred, origin)[0].item() - 9.0309) < 1e-3
    # last, smaller batch of the same loader: one sample = 3 canvases
    out1 = {"pred": out["pred"][:1], "origin_image": out["origin_image"][:1]}
    p1, _, gv1 = evalglue.compose_prediction_multiview(out1, mask[:3], batch_size=2, global_view_num=gv)
    assert gv1 == 3 and torch.equal(p1, pred[:1])
    # square views (no concat_target): the mask of view 0 applies as it is; area down-sampling 2 -> 1 averages the pasted tile
    m2 = torch.zeros(2 * 2, h, h, 1)
    m2[0, 0, 0] = 1
    out2 = {"pred": torch.full((2, 3, h, h), 1.0), "origin_image": torch.full((2, 3, h, h), 0.0)}
    p2, o2, _ = evalglue.compose_prediction_multiview(out2, m2, batch_size=2, test_size=2, metric_size=1)
    assert p2.shape == (2, 3, 1, 1) and abs(p2[0, 0, 0, 0].item() - 0.25) < 1e-7 and p2[1].abs().max().item() == 0.0 and o2.abs().max().item() == 0.0


def test_lpips_alex_restatement_structure_and_properties():
    """LPIPS(alex) of the harness (test_inpainting.py:159): `evalglue.LPIPSAlex` against a sequential restatement written with the
    lpips package's own state-dict key spelling (`net.slice{k}.{idx}.*`, `lin{k}.model.1.weight`), on random weights --
    PARITY-UNPINNED (the lpips package and its weights are absent here): the check is that the module computes the published
    formula from whichever key spelling it is given, is symmetric, zero on identical inputs, and refuses to run without weights."""
    import torch.nn.functional as F
    from leftrefill_amd.evalglue import LPIPSAlex
    g = torch.Generator().manual_seed(3)
    shapes = LPIPSAlex.CONVS
    idx = LPIPSAlex.FEATURE_INDEX
    sd_pkg, sd_tv, sd_lin = {}, {}, {}
    for k, (ci, co, ks, st, pd) in enumerate(shapes):
        w = torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5
        b = 0.1 * torch.randn(co, generator=g)
        lin = torch.rand(1, co, 1, 1, generator=g)
        sd_pkg[f"net.slice{k + 1}.{idx[k]}.weight"], sd_pkg[f"net.slice{k + 1}.{idx[k]}.bias"], sd_pkg[f"lin{k}.model.1.weight"] = w, b, lin
        sd_tv[f"features.{idx[k]}.weight"], sd_tv[f"features.{idx[k]}.bias"] = w, b
        sd_lin[f"lin{k}.model.1.weight"] = lin
    a = torch.rand(2, 3, 96, 80, generator=g) * 2 - 1
    b_img = (a + 0.3 * torch.randn(a.shape, generator=g)).clamp(-1, 1)

    def published(x0, x1):
        shift = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
        scale = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
        total = torch.zeros(x0.shape[0], 1, 1, 1)
        h0, h1 = (x0 - shift) / scale, (x1 - shift) / scale
        for k, (ci, co, ks, st, pd) in enumerate(shapes):
            if k in (1, 2):
                h0, h1 = F.max_pool2d(h0, kernel_size=3, stride=2), F.max_pool2d(h1, kernel_size=3, stride=2)
            w, bb = sd_pkg[f"net.slice{k + 1}.{idx[k]}.weight"], sd_pkg[f"net.slice{k + 1}.{idx[k]}.bias"]
            h0, h1 = F.relu(F.conv2d(h0, w, bb, st, pd)), F.relu(F.conv2d(h1, w, bb, st, pd))
            n0 = h0 / (torch.sqrt(torch.sum(h0 ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = h1 / (torch.sqrt(torch.sum(h1 ** 2, dim=1, keepdim=True)) + 1e-10)
            total = total + F.conv2d((n0 - n1) ** 2, sd_pkg[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
        return total

    want = published(a, b_img)
    m1 = LPIPSAlex().load_weights(sd_pkg)
    m2 = LPIPSAlex().load_weights(sd_tv, sd_lin)
    d1, d2 = m1(a, b_img), m2(a, b_img)
    assert d1.shape == (2, 1, 1, 1)
    assert torch.allclose(d1, want, rtol=1e-5, atol=1e-7) and torch.equal(d1, d2)
    assert torch.allclose(m1(b_img, a), d1, rtol=1e-6) and (d1 > 0).all()
    assert torch.equal(m1(a, a), torch.zeros(2, 1, 1, 1))
    with pytest.raises(RuntimeError, match="no weights"):
        LPIPSAlex()(a, b_img)
    with pytest.raises(KeyError, match="lin4.model.1.weight"):
        LPIPSAlex().load_weights(sd_tv, {k: v for k, v in sd_lin.items() if not k.startswith("lin4")})
