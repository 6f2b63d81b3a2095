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
    view_num - 1; every transformer block exchanges the reference halves (all_gather_into_tensor) and rank 0's target half
    (broadcast) over RCCL before the re-arranged self-attention (leftrefill_amd.dist.mv_gather_sequence).
    `mv_shard_graph = True` captures the whole step INCLUDING those collectives into the hipGraph (every rank captures and
    replays in lockstep; needs the `nccl` backend -- PyTorch records RCCL kernels like any other stream work); the default
    launches the sharded step eagerly."""
    st_cls = MultiViewSpatialTransformer
    mv_shard = False
    mv_shard_graph = False

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        from leftrefill_amd import engine
        if not self.mv_shard:
            return super().forward(x, timesteps, context, y, **kwargs)
        import torch.distributed as tdist
        from leftrefill_amd import dist as lrd
        graph_ok = self.mv_shard_graph and self.use_hip_graph and (
            (tdist.is_available() and tdist.is_initialized() and tdist.get_backend() == "nccl") or lrd._sim_world() > 0)
        prev_graph, prev_flag = self.use_hip_graph, engine.MV_SHARDED
        self.use_hip_graph, engine.MV_SHARDED = graph_ok, True
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
