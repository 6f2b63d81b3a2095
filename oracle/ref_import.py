"""Import the real reference (/root/reference) on CPU -- authoring container only.

Used by oracle/make_golden.py to produce tests/golden/*.npz.  Never runs on the
GPU box (the reference does not travel) and is never imported by the product.

The reference needs four packages this image lacks (torchvision, omegaconf,
pytorch_lightning, cv2); none of them is on the hot path, so they are stubbed
(SURVEY.md section 8c).  One behavioural patch: DDIMSampler.register_buffer
hard-codes `.to("cuda")` (ldm/models/diffusion/ddim.py:17-21).
"""
import os
import sys
import types

REF = os.environ.get("LEFTREFILL_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "ldm"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
        tv.transforms = _stub("torchvision.transforms")
    if "omegaconf" not in sys.modules:
        class ListConfig(list):
            pass

        class OmegaConf:
            @staticmethod
            def load(*a, **k):
                raise RuntimeError("omegaconf stub")

        oc = _stub("omegaconf", OmegaConf=OmegaConf, ListConfig=ListConfig)
        oc.listconfig = _stub("omegaconf.listconfig", ListConfig=ListConfig)
    if "pytorch_lightning" not in sys.modules:
        class LightningModule(torch.nn.Module):
            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")

            def log(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass

        pl = _stub("pytorch_lightning", LightningModule=LightningModule)
        pl.utilities = _stub("pytorch_lightning.utilities")
        pl.utilities.distributed = _stub("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    if "cv2" not in sys.modules:
        _stub("cv2")


def import_reference():
    """Returns a namespace with the reference classes used for goldens."""
    if not available():
        raise RuntimeError(f"reference not found at {REF}")
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # make sure we get the reference's `ldm`, not a drop-in one
    for k in list(sys.modules):
        if k == "ldm" or k.startswith("ldm."):
            del sys.modules[k]
    from ldm.modules.diffusionmodules import openaimodel, util as dutil
    from ldm.modules import attention
    from ldm.models.diffusion import ddim, ddpm
    from ldm.modules.diffusionmodules import multiview_unet
    from ldm.modules import multiview_attention

    ddim.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    ns = types.SimpleNamespace(openaimodel=openaimodel, attention=attention, dutil=dutil, ddim=ddim, ddpm=ddpm,
                               multiview_unet=multiview_unet, multiview_attention=multiview_attention)
    return ns
