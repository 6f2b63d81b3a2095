"""Drop-in `ldm.*` / `inpainting_ldm.*` / `dataloaders.*` packages with LeftRefill's module / operator API on the HIP kernels.

    import leftrefill_amd.dropin as dropin; dropin.install()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel      # same ctor kwargs / state_dict keys
    from inpainting_ldm.model import create_model, load_state_dict      # what test_inpainting.py imports

or simply put this directory first on PYTHONPATH.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def install():
    """Make `ldm`, `inpainting_ldm` and `dataloaders` resolve to this directory (idempotent)."""
    if sys.path[0:1] != [ROOT]:
        if ROOT in sys.path:
            sys.path.remove(ROOT)
        sys.path.insert(0, ROOT)
    for name in list(sys.modules):
        if name in ("ldm", "inpainting_ldm", "dataloaders") or name.startswith(("ldm.", "inpainting_ldm.", "dataloaders.")):
            mod = sys.modules[name]
            f = getattr(mod, "__file__", None) or ""
            if not f.startswith(ROOT):
                del sys.modules[name]
    return ROOT
