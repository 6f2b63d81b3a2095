"""Drop-in `dataloaders` package: only `test_dataset` is replaced.

The reference's `dataloaders` is a namespace directory that also holds inpainting_dataset, inpainting_crossview_dataset and
obj_nvs_dataset (training / multi-view entry points import them).  A regular package would shadow those, so this package
appends every other `dataloaders` directory found on sys.path to its search path: `dataloaders.test_dataset` resolves here,
everything else still resolves to the reference's files.
"""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in list(sys.path):
    _d = os.path.join(_p or os.getcwd(), "dataloaders")
    if os.path.isdir(_d) and os.path.abspath(_d) != _here and _d not in __path__:
        __path__.append(_d)
