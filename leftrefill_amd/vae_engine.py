"""KL-VAE encoder / decoder on the HIP kernels ("the step either side of the path").

Same conv3x3 / 1x1 implicit GEMM, GroupNorm(+SiLU) and layout kernels as the UNet step; two additions:
  * the encoder's Downsample pads only bottom/right (`lr_gemm_args.asym`);
  * the single-head d = C (512) AttnBlock materialises its fp16 logits like the reference's autocast path:
    logits = q k^T (GEMM, wt = k), softmax rows (lr_softmax_rows_f16), out = p v (GEMM, wt = v^T).  v^T comes straight out of
    a GEMM with the roles swapped (rows = W_v, "weights" = the normalised tokens) and the v bias is added after p v --
    exact because every softmax row sums to 1.

Reference semantics (ldm/modules/diffusionmodules/model.py in the reference repo):
  ResnetBlock.forward 122-150, AttnBlock.forward 175-204, Upsample 60-66, Downsample 80-88,
  Encoder.forward 517-544, Decoder.forward 615-653; AutoencoderKL.encode / decode, ldm/models/autoencoder.py:82-91.
"""
import torch

from . import ops
from .engine import Act, PackedConv, PackedLinear, PackedNorm, conv, gn, linear

MAX_OPERAND_BYTES = (1 << 31) - 1     # 32-bit gather offsets in lr_gemm_conv_f16


class PackedVaeRes:
    def __init__(self, blk):
        self.n1, self.c1 = PackedNorm(blk.norm1), PackedConv(blk.conv1)
        self.n2, self.c2 = PackedNorm(blk.norm2), PackedConv(blk.conv2)
        self.short = None
        if blk.in_channels != blk.out_channels:
            self.short = PackedConv(blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut)


class PackedVaeAttn:
    def __init__(self, blk):
        self.norm = PackedNorm(blk.norm)
        self.q, self.k = PackedLinear(blk.q), PackedLinear(blk.k)
        w = blk.v.weight.detach()
        self.wv = w.reshape(w.shape[0], w.shape[1]).to(torch.float16).contiguous()   # rows of the v^T GEMM
        self.bv = blk.v.bias.detach().float().contiguous()
        self.proj = PackedLinear(blk.proj_out)
        self.C = w.shape[0]


def vae_resblock(a: Act, p: PackedVaeRes):
    h = conv(gn(a, p.n1, True), p.c1)
    h = gn(h, p.n2, True)
    skip = a.tok if p.short is None else conv(a, p.short).tok
    return conv(h, p.c2, resid=skip)


def vae_attn(a: Act, p: PackedVaeAttn):
    C, L = p.C, a.HW
    if L % 64:
        raise RuntimeError(f"VAE AttnBlock on HIP needs H*W % 64 == 0 (got {L})")
    h = gn(a, p.norm, False).tok
    q, k = linear(h, p.q), linear(h, p.k)
    o = torch.empty(a.N * L, C, device=h.device, dtype=torch.float16)
    for b in range(a.N):           # one image at a time: the logits are [L, L]
        rows = slice(b * L, (b + 1) * L)
        s = ops.gemm_conv(q[rows], k[rows], B=1, H=1, W=L, taps=1)                      # q k^T, model.py:185
        ops.softmax_rows(s, float(C) ** -0.5, out=s)                                     # 186-187
        vt = ops.gemm_conv(p.wv, h[rows], B=1, H=1, W=C, taps=1)                         # (W_v h^T) = v^T - b_v
        ops.gemm_conv(s, vt, B=1, H=1, W=L, taps=1, bias=p.bv, out=o[rows])              # p v (+ b_v), 192
    return Act(linear(o, p.proj, resid=a.tok), a.N, a.H, a.W)


class PackedEncoder:
    def __init__(self, enc, quant_conv):
        self.conv_in = PackedConv(enc.conv_in)
        self.levels = []
        for lvl in enc.down:
            blocks = [PackedVaeRes(b) for b in lvl.block]
            attns = [PackedVaeAttn(t) for t in lvl.attn]
            ds = None
            if hasattr(lvl, "downsample"):
                if not lvl.downsample.with_conv:
                    raise RuntimeError("avg-pool Downsample is not used by the LeftRefill VAE")
                ds = PackedConv(lvl.downsample.conv)
            self.levels.append((blocks, attns, ds))
        self.mid1, self.mid_attn, self.mid2 = (PackedVaeRes(enc.mid.block_1), PackedVaeAttn(enc.mid.attn_1),
                                               PackedVaeRes(enc.mid.block_2))
        self.norm_out = PackedNorm(enc.norm_out)
        self.conv_out = PackedConv(enc.conv_out)
        self.quant = PackedConv(quant_conv, cin_pad=self.conv_out.w.shape[0])
        self.moments = quant_conv.weight.shape[0]


class PackedDecoder:
    def __init__(self, dec, post_quant_conv):
        self.post_quant = PackedConv(post_quant_conv)
        self.conv_in = PackedConv(dec.conv_in, cin_pad=self.post_quant.w.shape[0])
        self.mid1, self.mid_attn, self.mid2 = (PackedVaeRes(dec.mid.block_1), PackedVaeAttn(dec.mid.attn_1),
                                               PackedVaeRes(dec.mid.block_2))
        self.levels = []
        for lvl in dec.up:            # index == resolution level; executed in reverse
            blocks = [PackedVaeRes(b) for b in lvl.block]
            attns = [PackedVaeAttn(t) for t in lvl.attn]
            us = None
            if hasattr(lvl, "upsample"):
                if not lvl.upsample.with_conv:
                    raise RuntimeError("conv-less Upsample is not used by the LeftRefill VAE")
                us = PackedConv(lvl.upsample.conv)
            self.levels.append((blocks, attns, us))
        self.norm_out = PackedNorm(dec.norm_out)
        self.conv_out = PackedConv(dec.conv_out)
        self.out_ch = dec.conv_out.weight.shape[0]
        self.widest = max(b.c1.w.shape[0] for blocks, _, _ in self.levels for b in blocks)
        self.tanh_out = dec.tanh_out
        if dec.give_pre_end:
            raise RuntimeError("give_pre_end is not used by the LeftRefill VAE")


def _chunks(n_images, bytes_per_image):
    per = max(1, MAX_OPERAND_BYTES // max(1, bytes_per_image))
    return [(i, min(n_images, i + per)) for i in range(0, n_images, per)]


def encode_moments(x, p: PackedEncoder):
    """x [N, 3, H, W] fp32 -> moments [N, 2*z, H/8, W/8] fp32 = quant_conv(encoder(x))."""
    N, _, H, W = x.shape
    widest = p.conv_in.w.shape[0]
    outs = []
    for i0, i1 in _chunks(N, H * W * max(widest, 64) * 2):
        a = Act(ops.nchw_to_nhwc(x[i0:i1], cpad=p.conv_in.w.shape[1] // 9), i1 - i0, H, W)
        a = conv(a, p.conv_in)
        for blocks, attns, ds in p.levels:
            for j, blk in enumerate(blocks):
                a = vae_resblock(a, blk)
                if attns:
                    a = vae_attn(a, attns[j])
            if ds is not None:
                a = conv(a, ds, asym=True)
        a = vae_resblock(vae_attn(vae_resblock(a, p.mid1), p.mid_attn), p.mid2)
        a = conv(gn(a, p.norm_out, True), p.conv_out)
        a = conv(a, p.quant)
        outs.append(ops.nhwc_to_nchw(a.tok, a.N, a.H, a.W, p.moments, torch.float32))
    return outs[0] if len(outs) == 1 else torch.cat(outs)


def decode(z, p: PackedDecoder):
    """z [N, 4, h, w] fp32 -> image [N, 3, 8h, 8w] fp32 = decoder(post_quant_conv(z))."""
    N, _, h, w = z.shape
    scale = 2 ** (len(p.levels) - 1)
    # widest full-resolution operand: the last Upsample conv reads/writes levels[1]'s width at full resolution
    full_c = max([b.c1.w.shape[1] // 9 for b in p.levels[0][0]] + [64])
    outs = []
    for i0, i1 in _chunks(N, h * w * scale * scale * full_c * 2):
        a = Act(ops.nchw_to_nhwc(z[i0:i1], cpad=p.post_quant.w.shape[1]), i1 - i0, h, w)
        a = conv(a, p.post_quant)
        a = conv(a, p.conv_in)
        a = vae_resblock(vae_attn(vae_resblock(a, p.mid1), p.mid_attn), p.mid2)
        for blocks, attns, us in reversed(p.levels):
            for j, blk in enumerate(blocks):
                a = vae_resblock(a, blk)
                if attns:
                    a = vae_attn(a, attns[j])
            if us is not None:
                a = conv(a, us, up=1)
        a = conv(gn(a, p.norm_out, True), p.conv_out)
        y = ops.nhwc_to_nchw(a.tok, a.N, a.H, a.W, p.out_ch, torch.float32)
        outs.append(torch.tanh(y) if p.tanh_out else y)
    return outs[0] if len(outs) == 1 else torch.cat(outs)
