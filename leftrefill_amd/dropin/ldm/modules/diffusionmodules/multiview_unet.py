"""`ldm.modules.diffusionmodules.multiview_unet.MultiViewUnetModel` (reference multiview_unet.py:33-411).

The reference class is a copy of UNetModel that builds MultiViewSpatialTransformer blocks and takes three extra
kwargs (view_num / concat_target / no_rearrange_selfattn, lines 213-217, 270-274, 324-328).  Here it is a thin
subclass: same state-dict keys, same forward; the batch is the flattened '(b v)' canvas/view batch.
"""
from ldm.modules.diffusionmodules.openaimodel import *  # noqa: F401,F403  (the reference star-imports, line 13)
from ldm.modules.diffusionmodules.openaimodel import UNetModel
from ldm.modules.multiview_attention import MultiViewSpatialTransformer


class MultiViewUnetModel(UNetModel):
    st_cls = MultiViewSpatialTransformer

    def __init__(self, *args, view_num=4, concat_target=False, no_rearrange_selfattn=False, **kwargs):
        self.st_kwargs = dict(view_num=view_num, concat_target=concat_target,
                              no_rearrange_selfattn=no_rearrange_selfattn)
        self.view_num = view_num
        self.concat_target = concat_target
        super().__init__(*args, **kwargs)
