"""`ldm.modules.multiview_attention` on the MI355X kernels (reference ldm/modules/multiview_attention.py).

MultiViewBasicTransformerBlock (394-468) = BasicTransformerBlock whose *self*-attention runs over the tokens of all
views of one sample:
  concat_target=True : from V-1 stitched canvases [ref_i | target] build [target(canvas 0), ref_0 .. ref_{V-2}]
                       (view_num * s^2 tokens), attend, write target back to EVERY canvas (lines 440-446, 456-460)
                       -> lr_mv_gather / lr_mv_scatter around the fused attention;
  concat_target=False: '(b v) hw c -> b (v hw) c' (448, 462) is a free reshape of the token-major layout: the
                       attention kernel is simply launched with batch b and sequence v*hw.
MultiViewSpatialTransformer (516-606) only forwards view_num / concat_target / no_rearrange_selfattn to its blocks.
"""
from ldm.modules.attention import (BasicTransformerBlock, CrossAttention, FeedForward, GEGLU,  # noqa: F401
                                   MemoryEfficientCrossAttention, Normalize, SpatialTransformer)


class MultiViewBasicTransformerBlock(BasicTransformerBlock):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, **kwargs):
        super().__init__(dim, n_heads, d_head, dropout=dropout, context_dim=context_dim, gated_ff=gated_ff,
                         checkpoint=checkpoint, disable_self_attn=disable_self_attn)
        if disable_self_attn:
            raise ValueError("The model should not disable self attention as designed.")
        self.view_num = kwargs.get("view_num", 4)
        self.concat_target = kwargs.get("concat_target", False)
        self.no_rearrange_selfattn = kwargs.get("no_rearrange_selfattn", False)


class MultiViewSpatialTransformer(SpatialTransformer):
    block_cls = MultiViewBasicTransformerBlock

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True, **kwargs):
        super().__init__(in_channels, n_heads, d_head, depth=depth, dropout=dropout, context_dim=context_dim,
                         disable_self_attn=disable_self_attn, use_linear=use_linear, use_checkpoint=use_checkpoint,
                         view_num=kwargs.get("view_num", 4), concat_target=kwargs.get("concat_target", False),
                         no_rearrange_selfattn=kwargs.get("no_rearrange_selfattn", False))
