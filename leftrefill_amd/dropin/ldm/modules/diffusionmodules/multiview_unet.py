"""`ldm.modules.diffusionmodules.multiview_unet.MultiViewUnetModel` (reference multiview_unet.py:33-411).

The reference class is a copy of UNetModel that builds MultiViewSpatialTransformer blocks and takes three extra
kwargs (view_num / concat_target / no_rearrange_selfattn, lines 213-217, 270-274, 324-328).  Here it is a thin
subclass: same state-dict keys, same forward; the batch is the flattened '(b v)' canvas/view batch.
"""
from ldm.modules.diffusionmodules.openaimodel import *  # noqa: F401,F403  (the reference star-imports, line 13)
from ldm.modules.diffusionmodules.openaimodel import UNetModel
from ldm.modules.multiview_attention import MultiViewSpatialTransformer


class MultiViewUnetModel(UNetModel):
    """`mv_shard = True` (attribute, not in the reference) runs ONE canvas per rank: torch.distributed world size must be
    view_num - 1; every transformer block all-gathers the canvases over RCCL before the re-arranged self-attention
    (leftrefill_amd.dist).  The step is then launched eagerly (collectives are not captured into the hipGraph)."""
    st_cls = MultiViewSpatialTransformer
    mv_shard = False

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        from leftrefill_amd import engine
        if not self.mv_shard:
            return super().forward(x, timesteps, context, y, **kwargs)
        prev_graph, prev_flag = self.use_hip_graph, engine.MV_SHARDED
        self.use_hip_graph, engine.MV_SHARDED = False, True
        try:
            return super().forward(x, timesteps, context, y, **kwargs)
        finally:
            self.use_hip_graph, engine.MV_SHARDED = prev_graph, prev_flag

    def __init__(self, *args, view_num=4, concat_target=False, no_rearrange_selfattn=False, **kwargs):
        self.st_kwargs = dict(view_num=view_num, concat_target=concat_target,
                              no_rearrange_selfattn=no_rearrange_selfattn)
        self.view_num = view_num
        self.concat_target = concat_target
        super().__init__(*args, **kwargs)
