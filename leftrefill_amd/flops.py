"""Algorithmic work model of the UNet step (2*MAC of conv / linear / attention matmuls).

Walks the *drop-in* UNetModel's module tree, so the numbers follow the model actually benchmarked.  For the shipped
SD2-inpainting config this reproduces the survey's figures: 1849.75 GFLOP per forward per batch element at latent
64x128 (373.56 at 32x64).
"""
import torch.nn as nn


def unet_flops(model, H, W, ctx_len=77):
    """Returns dict(gemm=..., attn=..., total=...) FLOPs for ONE batch element."""
    from ldm.modules.attention import SpatialTransformer
    from ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample

    gemm = 0.0
    attn = 0.0
    mc = model.model_channels
    ted = 4 * mc
    gemm += 2.0 * (mc * ted + ted * ted)
    h, w = H, W

    def walk(seq):
        nonlocal gemm, attn, h, w
        for layer in seq:
            if isinstance(layer, ResBlock):
                cin, cout = layer.channels, layer.out_channels
                gemm += 2.0 * h * w * 9 * (cin * cout + cout * cout) + 2.0 * ted * cout
                if cin != cout:
                    gemm += 2.0 * h * w * cin * cout
            elif isinstance(layer, SpatialTransformer):
                ch = layer.in_channels
                hw = h * w
                for blk in layer.transformer_blocks:
                    cdim = blk.attn2.to_k.weight.shape[1]
                    gemm += 2.0 * hw * ch * ch * 4 + 2.0 * hw * ch * ch * 2 + 2.0 * ctx_len * cdim * ch * 2
                    gemm += 2.0 * hw * ch * 8 * ch + 2.0 * hw * 4 * ch * ch
                    vn = getattr(blk, "view_num", None)
                    if vn is None:
                        attn += 4.0 * hw * hw * ch
                    elif blk.concat_target and not blk.no_rearrange_selfattn:
                        L = vn * (hw // 2)            # [target, ref_0 .. ref_{V-2}] joint sequence per V-1 canvases
                        attn += 4.0 * L * L * ch / (vn - 1)
                    else:
                        v = vn - 1 if blk.concat_target else vn
                        attn += 4.0 * hw * (v * hw) * ch
                    attn += 4.0 * hw * ctx_len * ch
                gemm += 2.0 * hw * ch * ch * 2
            elif isinstance(layer, Downsample):
                h, w = h // 2, w // 2
                gemm += 2.0 * h * w * 9 * layer.channels * layer.out_channels
            elif isinstance(layer, Upsample):
                h, w = h * 2, w * 2
                gemm += 2.0 * h * w * 9 * layer.channels * layer.out_channels
            elif isinstance(layer, nn.Conv2d):
                gemm += 2.0 * h * w * 9 * layer.in_channels * layer.out_channels

    for blk in model.input_blocks:
        walk(blk)
    walk(model.middle_block)
    for blk in model.output_blocks:
        walk(blk)
    gemm += 2.0 * h * w * 9 * mc * model.out_channels
    return {"gemm": gemm, "attn": attn, "total": gemm + attn}


def unet_train_flops(model, H, W, ctx_len=77):
    """FLOPs of ONE training sample with frozen weights (gradient only to the context): forward + input-gradient backward.
    Every conv / linear on the gradient path has one dgrad GEMM of the forward's size; attention backward = 5 products against the
    forward's 2 (S recomputed, dP, dV, dQ, dK).  The layers in front of the first cross-attention (input conv, first ResBlock,
    proj_in, the first self-attention with its projections) receive no gradient: nothing trainable sits upstream of them."""
    f = unet_flops(model, H, W, ctx_len)
    mc = model.model_channels
    hw = H * W
    cin = model.in_channels
    prefix_gemm = 2.0 * hw * (9 * cin * mc + 9 * 2 * mc * mc + mc * mc + 4 * mc * mc) + 2.0 * 4 * mc * mc
    prefix_attn = 4.0 * hw * hw * mc
    bwd_gemm = max(f["gemm"] - prefix_gemm, 0.0)
    bwd_attn = 2.5 * max(f["attn"] - prefix_attn, 0.0)
    return {"forward": f["total"], "backward": bwd_gemm + bwd_attn, "total": f["total"] + bwd_gemm + bwd_attn,
            "backward_gemm": bwd_gemm, "backward_attn": bwd_attn}
