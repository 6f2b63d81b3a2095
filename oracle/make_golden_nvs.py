"""Golden vectors of the reference's NVSUnetModel (authoring container only; writes tests/golden/nvs.npz).

inpainting_ldm/NVS_ldm.py does not import as shipped: it names a dataset module that is not in the repository
(dataloaders.novel_view_synthesis_dataset) and three packages this image lacks (skimage, torchmetrics, torchvision).  None of them
touches NVSUnetModel's arithmetic, so they are bound to empty stand-ins here and the reference class itself runs on CPU with the
name-keyed weights of oracle/golden_spec.nvs_unet_state: separator-token insertion / removal around every block and the c_input
add after the first input block (NVS_ldm.py:22-104).  Only the outputs are stored.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import golden_spec as G, ref_import  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    ns = ref_import.import_reference()      # reference `ldm` on sys.path, stubs for torchvision / omegaconf / pytorch_lightning / cv2
    sys.modules["torchvision"].transforms.functional = _stub("torchvision.transforms.functional")
    _stub("skimage")
    _stub("skimage.metrics", structural_similarity=None)
    _stub("torchmetrics")
    _stub("torchmetrics.functional", peak_signal_noise_ratio=None)
    _stub("dataloaders")
    _stub("dataloaders.novel_view_synthesis_dataset", NVS_DTUDataset=None, WarpNVS_DTUDataset=None)
    _stub("dataloaders.obj_nvs_dataset", NVS_OBJDataset=None)
    for k in list(sys.modules):
        if k == "inpainting_ldm" or k.startswith("inpainting_ldm."):
            del sys.modules[k]
    from inpainting_ldm.NVS_ldm import NVSUnetModel
    cfg = G.CONFIGS["FULL"]
    out = {}
    for case, use_sep, c_shape, N, H, W, ts in G.NVS_UNET_CASES:
        m = NVSUnetModel(use_sep=use_sep, **cfg.kwargs())
        missing, unexpected = m.load_state_dict(G.nvs_unet_state(use_sep), strict=True)
        assert not missing and not unexpected
        m.eval()
        x, t, ctx, c_input = G.nvs_unet_inputs(case, c_shape, N, H, W, ts)
        with torch.no_grad():
            y = m(x, t, context=ctx, c_input=None if c_input is None else c_input.clone())
        out[case] = y.numpy()
        print(case, tuple(y.shape), float(y.abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "nvs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
