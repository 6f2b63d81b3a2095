"""Functional execution engine of the UNet step on the HIP kernels.

Every function works on token-major fp16 activations (`tok` = [N*H*W, C]) and on *packed* weights
(leftrefill_amd.packing) held in small `Packed*` records that the drop-in nn.Modules build once from their
reference-shaped parameters.  No torch math runs on the hot path -- torch only allocates buffers.

Reference semantics implemented here (file:line in the reference repo):
  ResBlock._forward              ldm/modules/diffusionmodules/openaimodel.py:254-274
  Downsample / Upsample          openaimodel.py:90-159
  SpatialTransformer.forward     ldm/modules/attention.py:393-419
  BasicTransformerBlock._forward attention.py:279-283
  CrossAttention.forward         attention.py:165-196 / 218-250
  FeedForward / GEGLU            attention.py:51-78
  MultiViewBasicTransformerBlock._forward  ldm/modules/multiview_attention.py:431-468
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import packing
from . import train_ops as ops     # == leftrefill_amd.ops unless autograd is recording and an input requires grad
from .ops import plan_batch_scale

# 16-bit storage / MFMA operand type of activations and packed weights: float16 (the reference's autocast type, every
# inference config) or bfloat16 (BASELINE configs[4]).  Accumulation, statistics and the softmax stay fp32 in both.
_COMPUTE = [torch.float16]


def compute_dtype():
    return _COMPUTE[0]


class compute:
    """`with engine.compute(torch.bfloat16): ...` -- packing and activation conversions inside use that 16-bit type."""

    def __init__(self, dtype):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"compute dtype must be float16 or bfloat16, got {dtype}")
        self.dtype = dtype

    def __enter__(self):
        self.prev = _COMPUTE[0]
        _COMPUTE[0] = self.dtype

    def __exit__(self, *exc):
        _COMPUTE[0] = self.prev


@dataclass
class Act:
    """NHWC fp16 activation: tok [N*H*W, C] (+ optional second tensor = virtual channel concat)."""
    tok: torch.Tensor
    N: int
    H: int
    W: int
    tok2: Optional[torch.Tensor] = None
    gs: Optional[tuple] = None       # GroupNorm statistics of tok from its producer's epilogue: (per-channel partials, rows per
                                     # block, per-group partials | None, chunks per sample) -- see ops.gemm_conv(want_gn_stats=True)
    gs2: Optional[tuple] = None      # ... of tok2

    @property
    def HW(self):
        return self.H * self.W

    def materialize(self):
        if self.tok2 is None:
            return self.tok
        return torch.cat([self.tok, self.tok2], dim=1).contiguous()


def f32(p):
    return p.detach().float().contiguous()


# ---------------------------------------------------------------------------------------------------------------
# packed parameter records
# ---------------------------------------------------------------------------------------------------------------
class PackedConv:
    def __init__(self, conv, cin_pad=None):
        w = conv.weight.detach()
        self.cout = w.shape[0]
        self.taps = w.shape[2] * w.shape[3]
        self.w = packing.pack_conv(w, cin_pad=cin_pad, dtype=compute_dtype())
        if cin_pad == 16 and self.taps == 9:
            # 16-channel source (the UNet's input conv): k = tap * 16 + c, zero-padded from 144 to 192 = three K-steps of four taps
            # (lr_gemm_conv_f16's c16 gather)
            self.w = torch.cat([self.w, self.w.new_zeros(self.w.shape[0], 192 - self.w.shape[1])], dim=1).contiguous()
        self.b = packing.pack_bias(conv.bias.detach(), self.w.shape[0]) if conv.bias is not None else None
        self.stride = conv.stride[0]


class PackedLinear:
    """norm: the nn.LayerNorm that feeds this Linear -- a second, LayerNorm-folded copy of the weights is packed next
    to the plain one (`wf`, `bf`, `cs`, see packing.fold_layernorm) for the inference path."""

    def __init__(self, lin=None, weight=None, bias=None, norm=None):
        if lin is not None:
            weight, bias = lin.weight, lin.bias
        w = weight.detach()
        if w.dim() == 4:  # 1x1 conv used as a linear (use_linear_in_transformer=False)
            w = w.reshape(w.shape[0], w.shape[1])
        self.w = packing.pack_linear(w, compute_dtype())
        self.b = f32(bias) if bias is not None else None
        self.wf = None
        if norm is not None:
            self.wf, self.bf, self.cs = packing.fold_layernorm(w, None if bias is None else bias.detach(),
                                                               norm.weight.detach(), norm.bias.detach(), compute_dtype())
            self.eps = float(norm.eps)


class PackedNorm:
    def __init__(self, norm):
        self.g = f32(norm.weight)
        self.b = f32(norm.bias)
        self.eps = float(norm.eps)


class PackedRes:
    def __init__(self, blk):
        self.n1 = PackedNorm(blk.in_layers[0])
        self.c1 = PackedConv(blk.in_layers[2])
        self.emb = PackedLinear(blk.emb_layers[1])
        self.n2 = PackedNorm(blk.out_layers[0])
        self.c2 = PackedConv(blk.out_layers[3])
        self.skip = None
        self._c2s = None      # built on first inference use (a training step re-packs every time the weights change and never
        self._c2s_ok = False  # runs the fused form -- the concatenated copy would be rebuilt per optimizer step for nothing)
        sk = blk.skip_connection
        if isinstance(sk, torch.nn.Conv2d):
            self.skip = PackedConv(sk)
            # `skip_connection(x) + h` (openaimodel.py:274) as extra K-steps of the last conv: weights [N][9 C | C_in], summed biases
            self._c2s_ok = (sk.kernel_size == (1, 1) and self.c2.taps == 9 and self.c2.stride == 1
                            and self.skip.w.shape[0] == self.c2.w.shape[0])
        self.cout = self.c1.cout

    @property
    def c2s(self):
        if self._c2s is None and self._c2s_ok:
            self._c2s = FusedSkipConv(self.c2, self.skip)
        return self._c2s


class FusedSkipConv:
    """The ResBlock's last 3x3 conv with its 1x1 skip_connection appended along K (lr_gemm_args.skip1): one accumulation, one rounding."""

    def __init__(self, c2, skip):
        self.w = torch.cat([c2.w, skip.w], dim=1).contiguous()
        b2 = c2.b if c2.b is not None else torch.zeros(c2.w.shape[0], device=c2.w.device)
        self.b = (b2 + skip.b) if skip.b is not None else c2.b
        self.cout, self.taps, self.stride = c2.cout, 9, 1


class PackedAttn:
    """attn1: fused [q;k;v] projection; attn2: q projection + fused [k;v] projection of the context."""

    def __init__(self, attn, is_self, norm=None):
        self.heads = attn.heads
        self.is_self = is_self
        if is_self:
            self.qkv = PackedLinear(weight=torch.cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], 0), norm=norm)
        else:
            self.q = PackedLinear(weight=attn.to_q.weight, norm=norm)
            self.kv = PackedLinear(weight=torch.cat([attn.to_k.weight, attn.to_v.weight], 0))
        self.out = PackedLinear(attn.to_out[0])
        self.xk = None
        if not is_self and norm is not None and attn.to_q.weight.shape[0] in ops.XATTN_WIDTHS:
            # operands of the fused cross-attention block (lr_xattn_block_f16): to_k rows / to_out columns in its k-slot order
            self.xk, self.xwo = packing.pack_xattn(attn.to_k.weight.detach(), attn.to_out[0].weight.detach(), compute_dtype())
            # LayerNorm-folded to_q with its columns in k-slot order: the variant that also runs the self-attention's out-projection
            # hands the normalised rows to the q projection in accumulator order
            self.xq_pi = self.q.wf[:, packing.xattn_perm(self.q.wf.shape[1], self.q.wf.device)].contiguous()
        self.dim_head = attn.to_q.weight.shape[0] // attn.heads
        if self.dim_head != 64:
            raise RuntimeError(f"attention kernel is specialised for d_head=64 (got {self.dim_head})")


class PackedTBlock:
    def __init__(self, blk):
        if getattr(blk, "disable_self_attn", False):
            raise RuntimeError("disable_self_attn=True is not used by LeftRefill configs and is unsupported")
        self.attn1 = PackedAttn(blk.attn1, True, blk.norm1)
        self.attn2 = PackedAttn(blk.attn2, False, blk.norm2)
        self.n1, self.n2, self.n3 = PackedNorm(blk.norm1), PackedNorm(blk.norm2), PackedNorm(blk.norm3)
        proj = blk.ff.net[0].proj
        self.geglu_w, self.geglu_b = packing.pack_geglu(proj.weight.detach(), proj.bias.detach(), compute_dtype())
        # LayerNorm(norm3)-folded copy of the GEGLU projection (rows in the same interleaved order)
        wf, bf, cs = packing.fold_layernorm(proj.weight.detach(), proj.bias.detach(), blk.norm3.weight.detach(),
                                            blk.norm3.bias.detach(), compute_dtype())
        perm = packing.geglu_perm(proj.weight.shape[0] // 2, proj.weight.device)
        self.geglu_wf, self.geglu_bf, self.geglu_cs = wf[perm].contiguous(), bf[perm].contiguous(), cs[perm].contiguous()
        self.ff2 = PackedLinear(blk.ff.net[2])
        self.ff2_x = None
        if self.ff2.w.shape[0] == ops.FFN_C and self.ff2.w.shape[1] % 64 == 0:
            # second Linear with its columns in the k-slot order of the fused feed-forward block (lr_ffn_block_f16)
            self.ff2_x = packing.pack_pieces(blk.ff.net[2].weight.detach(), compute_dtype())
        # multi-view attributes (None for the single-view block)
        self.kv_slot = None   # index into the per-context K/V projection cache (set by UNetModel.prepare)
        self.view_num = getattr(blk, "view_num", None)
        self.concat_target = getattr(blk, "concat_target", False)
        self.no_rearrange = getattr(blk, "no_rearrange_selfattn", False)
        if self.concat_target and self.no_rearrange:
            # the reference applies its forward rearrange twice on this branch (multiview_attention.py:437-438, 452-453): it cannot run
            # there (shape error unless b % (view_num - 1) == 0, then a context batch mismatch); no config uses it -- see oracle/unet_ref.py
            raise NotImplementedError("no_rearrange_selfattn=True with concat_target=True is unusable in the reference and not implemented")


class PackedST:
    def __init__(self, st):
        self.norm = PackedNorm(st.norm)
        self.proj_in = PackedLinear(st.proj_in)
        self.blocks = [PackedTBlock(b) for b in st.transformer_blocks]
        self.proj_out = PackedLinear(st.proj_out)
        self.proj_out_x = None
        if self.proj_out.w.shape == (ops.FFN_C, ops.FFN_C) and self.proj_out.b is not None:
            w_ = st.proj_out.weight.detach()
            self.proj_out_x = packing.pack_pieces(w_.reshape(w_.shape[0], w_.shape[1]), compute_dtype())      # fused behind the last block's feed-forward
        # proj_out composed with the last block's second feed-forward Linear (attention.py:75-77, 282, 412-419): two linear maps in a row,
        #   proj_out(ff2(g) + b2 + x) + bp + x_in = (Wp W2) g + Wp x + (Wp b2 + bp) + x_in,
        # run as ONE GEMM over [g | x] (lr_gemm_args.skip1 with taps == 1) where no fused feed-forward kernel exists (C > 320): the product
        # Wp W2 is formed in fp32 and rounded once, x3 = ff(..) + x is never rounded or written, a K = C launch per SpatialTransformer is gone
        # Formed on first inference use: ~110 GFLOP of host fp64 matmuls for an SD-size UNet, inference-only, and a training
        # step re-packs whenever the weights change.
        self._ff_proj = None
        self._ff_src = None
        ff2 = st.transformer_blocks[-1].ff.net[2]
        wp = st.proj_out.weight
        if wp.shape[0] == wp.shape[1] == ff2.weight.shape[0] and wp.shape[1] % 64 == 0:      # (C = 320 too: the split feed-forward path, ops.FFN_SPLIT)
            self._ff_src = (st.proj_out, ff2, compute_dtype())

    def _build_ff_proj(self):
        if self._ff_proj is None and self._ff_src is not None:
            # host fp64 matmuls and device <-> host copies: never inside a stream capture (UNetModel.finalize_inference runs this up front)
            assert not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()), "PackedST.ff_proj_w built inside a graph capture: call UNetModel.finalize_inference() first"
            proj_out, ff2, dt = self._ff_src
            wp = proj_out.weight.detach().float()
            wp = wp.reshape(wp.shape[0], wp.shape[1])
            # (in fp64 on the host: no GPU BLAS on any path of this package, and no dependence on a summation order)
            dev_ = wp.device
            wp64, w2_64 = wp.double().cpu(), ff2.weight.detach().double().cpu()
            w = torch.cat([(wp64 @ w2_64).float(), wp64.float()], dim=1).to(device=dev_, dtype=dt).contiguous()
            b2 = ff2.bias.detach().double().cpu() if ff2.bias is not None else torch.zeros(w2_64.shape[0], dtype=torch.float64)
            bp = proj_out.bias.detach().double().cpu() if proj_out.bias is not None else torch.zeros(wp64.shape[0], dtype=torch.float64)
            self._ff_proj = (w, (wp64 @ b2 + bp).float().to(dev_).contiguous())
        return self._ff_proj

    @property
    def ff_proj_w(self):
        p_ = self._build_ff_proj()
        return None if p_ is None else p_[0]

    @property
    def ff_proj_b(self):
        p_ = self._build_ff_proj()
        return None if p_ is None else p_[1]


# ---------------------------------------------------------------------------------------------------------------
# functional blocks
# ---------------------------------------------------------------------------------------------------------------
def linear(x, pl: PackedLinear, resid=None, M=None, want_stats=False):
    M = x.shape[0] if M is None else M
    return ops.gemm_conv(x, pl.w, B=1, H=1, W=M, taps=1, bias=pl.b, resid=resid, want_stats=want_stats)


def fold_ok(x):
    """The LayerNorm-folded GEMMs are inference kernels (no backward): use them unless autograd needs the LayerNorm."""
    return LN_FOLD and not (torch.is_grad_enabled() and x.requires_grad)


def ln_linear(x, st, pn: PackedNorm, pl: PackedLinear, wide=None):
    """Linear(LayerNorm(x)).  st: per-row (sum, sumsq) partials of x from the GEMM that produced it, or None.
    With st the LayerNorm is folded into the GEMM (no normalised tensor is written); otherwise the LayerNorm kernel runs.
    wide: "qkv" | "q" -- the kind of projection (ops.ROWLIN_WIDTHS) the row-resident kernel may take at this width."""
    if wide and pl.wf is not None and fold_ok(x) and ops.rowlin_ok(x.shape[0], x.shape[1], pl.wf.shape[0], wide):
        return ops.rowlin(x, pl.wf, pl.bf, eps=pl.eps)
    if st is not None and pl.wf is not None and fold_ok(x):
        return ops.gemm_conv(x, pl.wf, B=1, H=1, W=x.shape[0], taps=1, bias=pl.bf, ln=(st, pl.eps, pl.cs))
    return linear(ops.layer_norm(x, pn.g, pn.b, pn.eps), pl)


# GroupNorm statistics out of the producing GEMM's epilogue (inference path); LEFTREFILL_GN_FUSE=0 runs the statistics pass.
GN_FUSE = __import__("os").environ.get("LEFTREFILL_GN_FUSE", "1") != "0"
# a ResBlock's 1x1 skip_connection as extra K-steps of its last conv (lr_gemm_args.skip1); 0 = separate GEMM + residual epilogue
SKIP_FUSED = __import__("os").environ.get("LEFTREFILL_SKIP_FUSED", "1") != "0"
# SpatialTransformer.proj_out composed with the last feed-forward Linear into one GEMM (levels without the fused feed-forward kernel)
FF_PROJ = __import__("os").environ.get("LEFTREFILL_FF_PROJ", "1") != "0"


def gn_fuse_ok(x):
    """Inference-only GroupNorm fusions (no backward): the skip-fused conv, the row-resident entries, the one-launch `out` block."""
    return GN_FUSE and not (torch.is_grad_enabled() and x.requires_grad)


# differentiable forward (round 6): GroupNorm statistics from the producers' epilogues; LEFTREFILL_TRAIN_GN_STATS=0 runs the statistics pass
TRAIN_GN_STATS = __import__("os").environ.get("LEFTREFILL_TRAIN_GN_STATS", "1") != "0"
# ... and the residual branches' gradients added inside the LayerNorm / GroupNorm backward kernels; LEFTREFILL_TRAIN_FORK=0: autograd's fan-in adds
TRAIN_FORK = __import__("os").environ.get("LEFTREFILL_TRAIN_FORK", "1") != "0"


# ... and the skip-extended last conv of a width-changing ResBlock in the differentiable forward; LEFTREFILL_TRAIN_SKIP_FUSED=0: two GEMMs + residual
TRAIN_SKIP_FUSED = __import__("os").environ.get("LEFTREFILL_TRAIN_SKIP_FUSED", "1") != "0"


def gn_stats_ok(x):
    """Producer-epilogue GroupNorm statistics: also in the differentiable forward (round 6) -- no gradient flows through the sums, the
    consuming GroupNorm's backward re-derives mean / rstd from them (train_ops._GroupNorm)."""
    return GN_FUSE and (TRAIN_GN_STATS or not torch.is_grad_enabled())


def training(x):
    return torch.is_grad_enabled() and x.requires_grad


def forking(x):
    return TRAIN_FORK and training(x)


def conv(act: Act, pc: PackedConv, rowvec=None, resid=None, up=0, asym=False, gn_stats=False, skip=None, skip_parts=None):
    """3x3 pad-1 conv (stride 1|2, optional nearest-2x upsample; asym: pad bottom/right only) or 1x1 conv over an Act.
    gn_stats: the output feeds a GroupNorm -- let the epilogue produce its statistics (Act.gs).
    skip = (tok, tok2 | None): pointwise K extension (pc is a FusedSkipConv), see ops.gemm_conv."""
    if pc.taps == 9:
        if up:
            H, W = act.H * 2, act.W * 2
        elif pc.stride == 2:
            H, W = act.H // 2, act.W // 2
        else:
            H, W = act.H, act.W
    else:
        H, W = act.H, act.W
    want = gn_stats and gn_stats_ok(act.tok) and pc.cout == pc.w.shape[0]
    extra = {} if skip_parts is None else {"skip_parts": skip_parts}      # (training: the two layers of a skip-extended conv, for its backward)
    y = ops.gemm_conv(act.tok, pc.w, B=act.N, H=H, W=W, Hs=act.H, Ws=act.W, taps=pc.taps, stride=pc.stride, up=up,
                      asym=asym, x2=act.tok2, bias=pc.b, rowvec=rowvec, resid=resid, want_gn_stats=want, skip=skip, **extra)
    y, gs = y if want else (y, None)
    return Act(y, act.N, H, W, gs=gs)


def _gn_train_stats(act: Act):
    """How the differentiable GroupNorm gets its statistics (train_ops._GroupNorm `stats`): the producers' sums, else None = own pass."""
    HW = act.HW
    if not (act.gs is not None and HW % act.gs[1] == 0 and gn_stats_ok(act.tok)
            and (act.tok2 is None or (act.gs2 is not None and HW % act.gs2[1] == 0))):
        return None
    if act.tok2 is None and act.gs[2] is not None:
        return ("groups", act.gs[2], act.gs[3])
    return ("channels", act.gs, act.gs2)


def gn(act: Act, pn: PackedNorm, silu):
    """GroupNorm(32)(+SiLU) of the (virtually concatenated) activation; statistics from the producers when they came along."""
    HW = act.HW
    if training(act.tok) or (act.tok2 is not None and training(act.tok2)):
        return Act(ops.group_norm(act.tok, act.N, HW, pn.g, pn.b, pn.eps, silu, act.tok2, stats=_gn_train_stats(act)), act.N, act.H, act.W)
    fused = (act.gs is not None and HW % act.gs[1] == 0 and gn_fuse_ok(act.tok)
             and (act.tok2 is None or (act.gs2 is not None and HW % act.gs2[1] == 0)))
    if fused and act.tok2 is None and act.gs[2] is not None:
        # the producer reduced its tile to per-group sums: ONE launch (no finalize); a virtual concat keeps the per-channel partials
        # + finalize path, its groups can straddle the two sources
        y = ops.group_norm_groups(act.tok, act.N, HW, pn.g, pn.b, pn.eps, silu, act.gs[2], act.gs[3])
    elif fused:
        y = ops.group_norm_fused(act.tok, act.N, HW, pn.g, pn.b, pn.eps, silu, act.gs, act.tok2, act.gs2)
    else:
        y = ops.group_norm(act.tok, act.N, HW, pn.g, pn.b, pn.eps, silu, act.tok2)
    return Act(y, act.N, act.H, act.W)


def gn_fork(act: Act, pn: PackedNorm, silu):
    """Training: (GroupNorm(act), act') -- act' is `act` for the residual branch around the GroupNorm (the ResBlock's skip path, the
    SpatialTransformer's `+ x_in`): the gradient of that branch is then added inside the GroupNorm's backward kernel (lr_groupnorm_bwd_res)
    instead of by a separate fan-in add."""
    y, t1, t2 = ops.group_norm_fork(act.tok, act.N, act.HW, pn.g, pn.b, pn.eps, silu, act.tok2, stats=_gn_train_stats(act))
    return Act(y, act.N, act.H, act.W), Act(t1, act.N, act.H, act.W, tok2=t2)


def resblock(act: Act, pr: PackedRes, emb_out):
    """emb_out: [N, Cout] fp16 (row stride may exceed Cout) = emb_layers(emb), added to every pixel of sample n."""
    if forking(act.tok) or (act.tok2 is not None and forking(act.tok2)):
        h, act = gn_fork(act, pr.n1, True)
    else:
        h = gn(act, pr.n1, True)
    h = conv(h, pr.c1, rowvec=emb_out, gn_stats=True)
    h = gn(h, pr.n2, True)
    if pr.c2s is not None and SKIP_FUSED and gn_fuse_ok(act.tok) and gn_fuse_ok(h.tok) and (act.tok2 is None or gn_fuse_ok(act.tok2)):
        # inference: the 1x1 skip_connection of a width-changing block rides on the last conv's K loop (no separate GEMM, no residual pass)
        return conv(h, pr.c2s, gn_stats=True, skip=(act.tok, act.tok2))
    if (pr.c2s is not None and SKIP_FUSED and TRAIN_SKIP_FUSED and (training(h.tok) or training(act.tok) or (act.tok2 is not None and training(act.tok2)))
            and h.tok2 is None):
        # training (round 6): the same one-launch form; its backward runs the 3x3 and the pointwise input-gradient GEMMs on the layers' own weights
        return conv(h, pr.c2s, gn_stats=True, skip=(act.tok, act.tok2), skip_parts=(pr.c2.w, pr.skip.w))
    if pr.skip is not None:
        resid = conv(act, pr.skip).tok
    else:
        resid = act.materialize()
    return conv(h, pr.c2, resid=resid, gn_stats=True)


def attention_plain(x, ctx, pa: PackedAttn, B, L, Lc=None):
    """CrossAttention.forward on its own (attention.py:165-196): to_out(attention(x Wq, c Wk, c Wv)), c = x when self."""
    if pa.is_self:
        a = ops.attention_qkv(linear(x, pa.qkv), B, pa.heads, L, pa.dim_head ** -0.5)
    else:
        a = ops.attention_q_kv(linear(x, pa.q), linear(ctx, pa.kv), B, pa.heads, L, Lc, pa.dim_head ** -0.5)
    return linear(a, pa.out)


def self_attention(x, st, pn, pa: PackedAttn, B, L, want_stats=False, qkv=None, wide=None):
    """x + to_out(attention(LayerNorm(x) Wqkv)); st: row statistics of x (or None); qkv: the projection when its producer already
    made it (ops.stin_block); wide: single-view blocks let the row-resident kernel take the projection (the multi-view forms keep the
    LayerNorm-folded GEMM, whose arithmetic the sharded path reproduces bit for bit)."""
    if qkv is None and forking(x):
        n, x = ops.layer_norm_fork(x, pn.g, pn.b, pn.eps)      # the residual's gradient joins inside the LayerNorm backward kernel
        qkv = linear(n, pa.qkv)
    qkv = ln_linear(x, st, pn, pa.qkv, wide=wide) if qkv is None else qkv
    a = ops.attention_qkv(qkv, B, pa.heads, L, pa.dim_head ** -0.5)
    return linear(a, pa.out, resid=x, want_stats=want_stats)


def dup2(t):
    """[rows, ...] -> [2 * rows, ...]: the (uncond | cond) halves of a classifier-free-guidance batch that are still identical."""
    return torch.cat([t, t], dim=0)


def xattn_fused(x, pa: PackedAttn, B, L, Lc, kv):
    """Will cross_attention take the one-launch path (lr_xattn_block_f16)?  It normalises its rows itself: the producer of x then
    does not need to emit row statistics."""
    return (XATTN and kv is not None and len(kv) > 2 and fold_ok(x) and ops.xattn_ok(B * L, L, x.shape[1], pa.heads, Lc))


def ffn_fused(x, pt, rows=None, ctx=None):
    """Will _ffn take the one-launch path (lr_ffn_block_f16)?  rows: rows of the tensor the feed-forward will see (x may still hold
    half of a CFG batch when the decision is taken, see transformer_block); ctx: the block's context -- when it requires grad the
    feed-forward input will too (it sits behind the cross-attention), and the fused kernel has no backward."""
    rows = x.shape[0] if rows is None else rows
    if ctx is not None and torch.is_grad_enabled() and ctx.requires_grad:
        return False
    return FFN_FUSED and pt.ff2_x is not None and fold_ok(x) and ops.ffn_ok(rows, x.shape[1], pt.ff2_x.shape[0] * 64)


XATTN_PRE = __import__("os").environ.get("LEFTREFILL_XATTN_PRE", "1") != "0"


def self_then_cross_attention(x, st, pt, N, L, Lc, kv, want_stats, dup=False, qkv0=None):
    """attn1 and attn2 of a plain (single-view) block with the self-attention's out-projection fused into the cross-attention launch:
    LayerNorm-folded QKV GEMM -> flash attention -> ONE kernel for  x1 = a Wo1 + b + x;  x2 = x1 + to_out(attention(LN(x1) Wq, K, V)).
    dup: x holds the first N / 2 samples of a CFG batch whose halves are still identical: projection and self-attention run on them
    once (planned like the full batch), then a and x are duplicated for the N contexts."""
    pa1, pa2 = pt.attn1, pt.attn2
    if dup:
        with plan_batch_scale(2):
            qkv = ln_linear(x, st, pt.n1, pa1.qkv, wide="qkv") if qkv0 is None else qkv0
        a = dup2(ops.attention_qkv(qkv, N // 2, pa1.heads, L, pa1.dim_head ** -0.5))
        x = dup2(x)
    else:
        qkv = ln_linear(x, st, pt.n1, pa1.qkv, wide="qkv") if qkv0 is None else qkv0
        a = ops.attention_qkv(qkv, N, pa1.heads, L, pa1.dim_head ** -0.5)
    return ops.xattn_block(x, pa2.xq_pi, pa2.q.bf, kv[2], kv[3], pa2.xwo, pa2.out.b, HW=L, heads=pa2.heads, Lc=Lc, eps=pa2.q.eps,
                           scale=pa2.dim_head ** -0.5, want_stats=want_stats, pre=(a, pa1.out.w, pa1.out.b))


def cross_attention(x, st, pn, ctx, pa: PackedAttn, B, L, Lc, kv=None, want_stats=False, dup=False, fused=None):
    """x + to_out(attention(LayerNorm(x) Wq, ctx Wk, ctx Wv)).  kv: optional precomputed ([B*Lc, 2C] = ctx @ [Wk; Wv]^T,
    V^T in the attention kernel's layout) -- constant over the DDIM steps, see UNetModel._context_kv.
    dup: x holds only the first B / 2 samples (the two CFG halves are identical up to here): the query projection runs
    once, then x and q are duplicated for the B contexts."""
    if xattn_fused(x, pa, B, L, Lc, kv) if fused is None else fused:
        # one launch: LayerNorm + to_q + attention + to_out + residual (+ the row statistics of the next LayerNorm)
        if dup:
            x = dup2(x)
        return ops.xattn_block(x, pa.q.wf, pa.q.bf, kv[2], kv[3], pa.xwo, pa.out.b, HW=L, heads=pa.heads, Lc=Lc, eps=pa.q.eps,
                               scale=pa.dim_head ** -0.5, want_stats=want_stats)
    if dup:
        with plan_batch_scale(2):
            q = ln_linear(x, st, pn, pa.q, wide="q")
        q, x = dup2(q), dup2(x)
    elif forking(x):
        n, x = ops.layer_norm_fork(x, pn.g, pn.b, pn.eps)
        q = linear(n, pa.q)
    else:
        q = ln_linear(x, st, pn, pa.q, wide="q")
    vt = None
    if kv is None:
        kv = linear(ctx, pa.kv)
    else:
        kv, vt = kv[0], kv[1]
    a = ops.attention_q_kv(q, kv, B, pa.heads, L, Lc, pa.dim_head ** -0.5, vt=vt)
    return linear(a, pa.out, resid=x, want_stats=want_stats)


def transformer_block(x, ctx, pt: PackedTBlock, N, L, Lc, kv=None, st=None, want_stats=False, dup=False, post=None, qkv0=None, ffn_split=False):
    """x [N*L, C]; ctx [N*Lc, Dc].  attention.py:279-283 / multiview_attention.py:431-468.
    st: per-row statistics of x from its producer (enables the LayerNorm fold); returns (x, statistics of x | None).
    dup (single-view blocks only): x carries the first N / 2 samples of a CFG batch whose halves are identical; the
    self-attention and the cross-attention's query projection run on them once (see UNetModel.cfg_shared_prefix).
    qkv0 (single-view blocks): the fused q|k|v projection of LayerNorm(x) when the producer of x already made it (ops.stin_block)."""
    # The two fused-block decisions are taken ONCE per block: they steer which producers emit row statistics, so every
    # consumer below must see the same answer.  (Both are pure functions of shapes, switches and the autograd state.)
    use_xattn = xattn_fused(x, pt.attn2, N, L, Lc, kv)
    use_ffn = ffn_fused(x, pt, rows=N * L, ctx=ctx) and not ffn_split      # rows of the block's output: x may still hold half of a CFG batch here (`dup`)
    # ask the residual GEMMs for the row statistics the next LayerNorm fold needs (the fused blocks normalise their rows themselves)
    ws = fold_ok(x) and not use_xattn
    if pt.view_num is None and XATTN_PRE and use_xattn and pt.attn1.out.b is not None:
        ws = fold_ok(x) and not use_ffn
        x = self_then_cross_attention(x, st, pt, N, L, Lc, kv, ws, dup=dup, qkv0=qkv0)
        x, st = x if ws else (x, None)
        return _ffn(x, st, pt, want_stats, post, use_ffn)
    if pt.view_num is None and dup:
        with plan_batch_scale(2):
            x = self_attention(x, st, pt.n1, pt.attn1, N // 2, L, want_stats=ws, qkv=qkv0, wide="qkv")
    elif pt.view_num is None:
        x = self_attention(x, st, pt.n1, pt.attn1, N, L, want_stats=ws, qkv=qkv0, wide="qkv")
    elif pt.concat_target and not pt.no_rearrange and MV_SHARDED:
        x = _mv_sharded_self_attention(x, pt, N, L, st)
        ws = False
    elif pt.concat_target and not pt.no_rearrange:
        v = pt.view_num - 1
        b = N // v
        s = int(math.sqrt(L / 2))
        assert 2 * s * s == L and b * v == N, "concat_target needs square halves and batch = b*(view_num-1)"
        seq = ops.mv_gather(x, b, v, s)
        Ls = pt.view_num * s * s
        st_seq = None
        if st is not None and fold_ok(x) and MV_LN_FOLD:
            # the rows' LayerNorm statistics take the same re-arrangement (8 bytes x parts per row; torch index glue), so the fused
            # QKV projection of the sequence folds its LayerNorm like every other block -- and stays bit-identical to the sharded path
            g = st.reshape(b, v, s, 2 * s, st.shape[1] * 2)
            st_seq = torch.cat((g[:, 0:1, :, s:], g[:, :, :, :s]), dim=1).reshape(b * Ls, st.shape[1], 2).contiguous()
        seq = self_attention(seq, st_seq, pt.n1, pt.attn1, b, Ls)
        x = ops.mv_scatter(seq, b, v, s)
        ws = False
    else:
        v = pt.view_num - 1 if pt.concat_target else pt.view_num
        b = N // v
        assert b * v == N
        x = self_attention(x, st, pt.n1, pt.attn1, b, v * L, want_stats=ws)
    x, st = x if ws else (x, None)
    ws = fold_ok(x) and not use_ffn
    x = cross_attention(x, st, pt.n2, ctx, pt.attn2, N, L, Lc, kv, want_stats=ws, dup=dup, fused=use_xattn)
    x, st = x if ws else (x, None)
    return _ffn(x, st, pt, want_stats, post, use_ffn)


FFN_POST = __import__("os").environ.get("LEFTREFILL_FFN_POST", "1") != "0"


def _ffn(x, st, pt: PackedTBlock, want_stats, post=None, fused=None):
    """x + ff(LayerNorm(x)) (attention.py:282); st: row statistics of x (or None); returns (y, statistics of y | None).
    post = (proj_out pieces, bias, x_in, want_gn): when the fused kernel runs, SpatialTransformer.proj_out (+ x_in) rides behind it in
    the same launch and the result is ("post", out, GroupNorm statistics | None) instead."""
    fused = ffn_fused(x, pt) if fused is None else fused
    compose = post is not None and post[0] == "compose"
    if compose and (fused or (torch.is_grad_enabled() and x.requires_grad)):      # (the composed GEMM has no backward: training keeps two layers)
        compose, post = False, None
    if post is not None and not compose and FFN_POST and fused:
        pw, pb, x_in, want_gn, hw = post
        y = ops.ffn_block(x, pt.geglu_wf, pt.geglu_bf, pt.ff2_x, pt.ff2.b, eps=pt.n3.eps, post=(pw, pb, x_in), want_gn_stats=want_gn,
                          gn_hw=hw)
        return ("post",) + (y if want_gn else (y, None))
    if fused:
        # one launch: LayerNorm + GEGLU projection + gate + second Linear + residual; the hidden activation stays in registers
        ws = want_stats and fold_ok(x)
        y = ops.ffn_block(x, pt.geglu_wf, pt.geglu_bf, pt.ff2_x, pt.ff2.b, eps=pt.n3.eps, want_stats=ws)
        return y if ws else (y, None)
    if fold_ok(x) and ops.rowlin_ok(x.shape[0], x.shape[1], pt.geglu_wf.shape[0], "geglu"):
        # C = 640: LayerNorm + GEGLU projection + gate with the rows resident in registers (lr_rowlin_f16)
        g = ops.rowlin(x, pt.geglu_wf, pt.geglu_bf, eps=pt.n3.eps, geglu=True)
    elif st is not None:
        g = ops.gemm_conv(x, pt.geglu_wf, B=1, H=1, W=x.shape[0], taps=1, bias=pt.geglu_bf, geglu=True,
                          ln=(st, pt.n3.eps, pt.geglu_cs))
    else:
        n3, x = (ops.layer_norm_fork if TRAIN_FORK else (lambda *a_: (ops.layer_norm(*a_), x)))(x, pt.n3.g, pt.n3.b, pt.n3.eps)      # (training: the residual below takes the fork's x)
        g = ops.gemm_conv(n3, pt.geglu_w, B=1, H=1, W=n3.shape[0], taps=1, bias=pt.geglu_b, geglu=True)
    if compose:
        # SpatialTransformer.proj_out behind the block: (Wp W2) g + Wp x + b' + x_in in one GEMM over [g | x] (PackedST.ff_proj_w)
        _, wc, bc, x_in, hw = post
        y, gs = ops.gemm_conv(g, wc, B=1, H=1, W=g.shape[0], taps=1, bias=bc, resid=x_in, skip=(x, None), want_gn_stats=True, gn_hw=hw)
        return ("post", y, gs)
    ws = want_stats and fold_ok(x)
    y = linear(g, pt.ff2, resid=x, want_stats=ws)
    return y if ws else (y, None)


# Fused cross-attention block for the C = 320 level (needs the per-context K / V^T pack of UNetModel._context_kv);
# LEFTREFILL_XATTN=0 keeps the to_q -> attention -> to_out launches.
XATTN = __import__("os").environ.get("LEFTREFILL_XATTN", "1") != "0"
# Fused feed-forward block for the C = 320 level; LEFTREFILL_FFN_FUSED=0 keeps the GEGLU GEMM -> Linear GEMM launches.
FFN_FUSED = __import__("os").environ.get("LEFTREFILL_FFN_FUSED", "1") != "0"

# LayerNorm folded into the consuming GEMM (inference path); LEFTREFILL_LN_FOLD=0 runs the stand-alone LayerNorm kernel.
LN_FOLD = __import__("os").environ.get("LEFTREFILL_LN_FOLD", "1") != "0"

# LayerNorm fold in the re-arranged multi-view self-attention (fused and sharded forms); LEFTREFILL_MV_LN_FOLD=0: stand-alone LayerNorm
MV_LN_FOLD = __import__("os").environ.get("LEFTREFILL_MV_LN_FOLD", "1") != "0"
# One canvas per rank (torch.distributed world == view_num - 1): set by UNetModel when `mv_shard=True`.
MV_SHARDED = False


# target query rows split over the ranks of a sharded multi-view job (one all-gather of the new target rows per block) instead of
# replicated on every rank
MV_SPLIT_TARGET = __import__("os").environ.get("LEFTREFILL_MV_SPLIT_TARGET", "1") != "0"
# the glue of the sharded block through lr_row_copy (three launches per block); 0: the torch slice / cat form the CPU (gloo) tests pin
MV_ROW_COPY = __import__("os").environ.get("LEFTREFILL_MV_ROW_COPY", "1") != "0"


def _mv_sharded_self_attention(x, pt: PackedTBlock, N, L, st=None):
    """Re-arranged cross-view self-attention with the canvases of a sample spread over the ranks (leftrefill_amd.dist).
    x [N*L, C] = this rank's canvas for each of its N local samples.  ONE all-gather delivers every rank's canvas rows together with
    their LayerNorm partial sums (mv_exchange_canvases); K / V are built for the whole sequence [target of canvas 0, ref_0 ..
    ref_{v-1}], Q / attention / out-projection only for the rows this rank owns:
      * MV_SPLIT_TARGET (default): slice `rank` of the target rows + ref_rank; the new target slices are all-gathered behind the
        out-projection (second collective), so every rank writes the same target half;
      * else: [target, ref_rank] -- the target rows are replicated, bit-identical work on every rank, no second exchange.
    st: per-row (sum, sumsq) partials of x from its producer; with them the LayerNorm is folded into the K|V and Q projections."""
    from . import dist as lrd
    C = x.shape[1]
    s = int(math.sqrt(L / 2))
    assert 2 * s * s == L
    v = pt.view_num - 1
    world = lrd.mv_group_size()
    assert world == v, f"multi-view sharding needs world_size == view_num - 1 ({world} vs {v})"
    rank = lrd.mv_rank()
    s2 = s * s
    Ls = (v + 1) * s2
    pq = pt.attn1.qkv                                                         # rows [Wq; Wk; Wv] of the fused projection
    fold = st is not None and pq.wf is not None and fold_ok(x) and MV_LN_FOLD
    parts = st.shape[1] if fold else 0
    split = MV_SPLIT_TARGET and world > 1 and s2 % world == 0
    Lo = s2 // world + s2 if split else L
    if x.is_cuda and MV_ROW_COPY:
        # the glue as three launches of lr_row_copy with cached index tables (pack | unpack | write-back) around the collectives
        E = 2 * parts
        rb = 2 * C + 4 * E
        plan = lrd.mv_shard_plan(N, v, s, rank, split, x.device)
        send = torch.empty(N * L, rb, dtype=torch.uint8, device=x.device)
        jobs = [dict(src=x, dst=send, row_bytes=2 * C, n_rows=N * L)]
        if fold:
            jobs.append(dict(src=st.reshape(N * L, E), dst=send, dst_off=2 * C, row_bytes=4 * E, n_rows=N * L))
        ops.row_copy(jobs)
        recv = lrd.mv_all_gather_rows(send)                                   # [v N L, rb] bytes, canvas-major: ONE collective
        seq = torch.empty(N * Ls, C, dtype=x.dtype, device=x.device)
        own_x = torch.empty(N * Lo, C, dtype=x.dtype, device=x.device)
        jobs = [dict(src=recv, dst=seq, row_bytes=2 * C, n_rows=N * Ls, src_idx=plan["seq_src"]),
                dict(src=recv, dst=own_x, row_bytes=2 * C, n_rows=N * Lo, src_idx=plan["own_src"])]
        if fold:
            st_seq = torch.empty(N * Ls, parts, 2, dtype=torch.float32, device=x.device)
            own_st = torch.empty(N * Lo, parts, 2, dtype=torch.float32, device=x.device)
            jobs += [dict(src=recv, src_off=2 * C, dst=st_seq.reshape(N * Ls, E), row_bytes=4 * E, n_rows=N * Ls, src_idx=plan["seq_src"]),
                     dict(src=recv, src_off=2 * C, dst=own_st.reshape(N * Lo, E), row_bytes=4 * E, n_rows=N * Lo, src_idx=plan["own_src"])]
        ops.row_copy(jobs)
    else:
        x_all, st_all = lrd.mv_exchange_canvases(x.reshape(N, L, C), st.reshape(N, L, parts * 2) if fold else None)
        seq = lrd.mv_sequence_from_canvases(x_all, s).reshape(N * Ls, C)      # [N Ls, C] (one copy out of the receive buffer)
        own = (lambda t_: lrd.mv_own_rows_split(t_, rank, s, world)) if split else (lambda t_: lrd.mv_own_rows(t_, rank, s))
        own_x = own(seq.reshape(N, Ls, C)).reshape(N * Lo, C)
        if fold:
            st_seq = lrd.mv_sequence_from_canvases(st_all, s)                 # [N, Ls, parts * 2] fp32
            own_st = own(st_seq).reshape(N * Lo, parts, 2).contiguous()
            st_seq = st_seq.reshape(N * Ls, parts, 2)
    if fold:
        kv = ops.gemm_conv(seq, pq.wf[C:], B=1, H=1, W=N * Ls, taps=1, bias=pq.bf[C:], ln=(st_seq, pq.eps, pq.cs[C:]))
        q = ops.gemm_conv(own_x, pq.wf[:C], B=1, H=1, W=N * Lo, taps=1, bias=pq.bf[:C], ln=(own_st, pq.eps, pq.cs[:C]))
    else:
        n_seq = ops.layer_norm(seq, pt.n1.g, pt.n1.b, pt.n1.eps)
        kv = ops.gemm_conv(n_seq, pq.w[C:], B=1, H=1, W=N * Ls, taps=1)       # K | V for every row of the sequence
        n_own = ops.layer_norm(own_x, pt.n1.g, pt.n1.b, pt.n1.eps)            # (row-wise: the own rows' LayerNorm again, 1 / 4 of the sequence)
        q = ops.gemm_conv(n_own, pq.w[:C], B=1, H=1, W=N * Lo, taps=1)
    a = ops.attention(q, kv[:, :C], kv[:, C:], N, pt.attn1.heads, Lo, Ls, pt.attn1.dim_head ** -0.5)
    y = linear(a, pt.attn1.out, resid=own_x)                                  # rows [target' (slice), ref_rank']
    if not split:
        return ops.mv_scatter(y, N, 1, s)                                     # -> canvas [ref' | target']
    n_t = s2 // world
    tgt_all = lrd.mv_all_gather_rows(y.reshape(N, Lo, C)[:, :n_t].contiguous())          # [v N, n_t, C]: second collective
    if x.is_cuda and MV_ROW_COPY:
        canvas = torch.empty(N * L, C, dtype=x.dtype, device=x.device)
        ops.row_copy([dict(src=y, dst=canvas, row_bytes=2 * C, n_rows=N * s2, src_idx=plan["ref_src"], dst_idx=plan["ref_dst"]),
                      dict(src=tgt_all.reshape(world * N * n_t, C), dst=canvas, row_bytes=2 * C, n_rows=N * s2, src_idx=plan["tgt_src"],
                           dst_idx=plan["tgt_dst"])])
        return canvas
    tgt = tgt_all.reshape(world, N, n_t, C).permute(1, 0, 2, 3).reshape(N, s2, C)
    y = torch.cat([tgt, y.reshape(N, Lo, C)[:, n_t:]], dim=1).reshape(N * L, C)
    return ops.mv_scatter(y, N, 1, s)


# GroupNorm of the SpatialTransformer folded into proj_in through per-sample weights (levels where the activation is much larger
# than N copies of the weights).  OFF by default since round 5: its own A/B is noise (18.43 vs 18.41 ms per step, round 4) and the
# folded weights W gamma rstd carry no range guard for near-constant groups (rstd up to 1e3 at eps 1e-6) -- the default
# path is GroupNorm-apply -> proj_in, whose normalised activations are bounded.  LEFTREFILL_ST_GN_FOLD=1 enables the fold with a
# developer build of the library (round 6: lr_gn_fold_weights_f16 and its parity test are compiled under -DLR_DEV_VARIANTS only).
ST_GN_FOLD = __import__("os").environ.get("LEFTREFILL_ST_GN_FOLD", "0") != "0"


def st_gn_fold_ok(x_in, act: Act, gs_in, ps: PackedST):
    C = x_in.shape[1]
    return (ST_GN_FOLD and ops._lib.dev_variants() and gs_in is not None and gs_in[2] is not None and fold_ok(x_in) and gn_fuse_ok(x_in)
            and x_in.shape[0] >= 2 * act.N * C and act.HW % 256 == 0 and ps.proj_in.w.shape == (C, C))


# proj_in + LayerNorm + q|k|v projection of a SpatialTransformer's first block as one launch (level 0); LEFTREFILL_STIN=0 keeps the two GEMMs
STIN = __import__("os").environ.get("LEFTREFILL_STIN", "1") != "0"


# the SpatialTransformer's GroupNorm applied inside that launch (needs the producer's per-group partials); LEFTREFILL_STIN_GN=0: gn_apply pass
STIN_GN = __import__("os").environ.get("LEFTREFILL_STIN_GN", "1") != "0"


def stin_fused(h, ps: PackedST):
    """Will the entry of the SpatialTransformer take the one-launch path (lr_stin_block_f16)?  Single-view blocks only: a multi-view
    block projects the re-arranged sequence, not the rows of h."""
    if not (STIN and ps.blocks and ps.blocks[0].view_num is None and fold_ok(h) and ps.proj_in.b is not None):
        return False
    pq = ps.blocks[0].attn1.qkv
    return pq.wf is not None and ps.proj_in.w.shape == (h.shape[1], h.shape[1]) and ops.stin_ok(h.shape[0], h.shape[1], pq.wf.shape[0])


def st_dup_ok(ps: PackedST):
    return len(ps.blocks) > 0 and ps.blocks[0].view_num is None


def spatial_transformer(act: Act, ctx, Lc, ps: PackedST, kv_cache=None, dup=False):
    """dup: `act` holds the first half of a CFG batch whose halves are still identical (ctx has all 2 * act.N contexts);
    the result is the full batch."""
    x_in = act.materialize()
    gs_in = act.gs if act.tok2 is None else None
    qkv0 = None
    with plan_batch_scale(2 if dup else 1):
        if st_gn_fold_ok(x_in, act, gs_in, ps):
            # Normalize (GroupNorm(32, eps 1e-6, affine), attention.py:399-404) folded into proj_in: per-sample weights from the
            # producer's per-group sums; the GEMM reads the RAW x_in and the normalised tensor is never written
            ws = True
            wb, bb = ops.gn_fold_weights(gs_in[2], gs_in[3], act.N, act.HW, ps.norm.g, ps.norm.b, ps.norm.eps, ps.proj_in.w, ps.proj_in.b)
            h = ops.gemm_conv(x_in, wb, B=act.N, H=1, W=act.HW, taps=1, bias=bb, per_sample=True, want_stats=True)
        elif (STIN_GN and stin_fused(x_in, ps) and gs_in is not None and gs_in[2] is not None and gn_fuse_ok(x_in)
              and act.HW % ops.STIN_ROWS == 0):
            # level 0: GroupNorm + proj_in + LayerNorm + q|k|v in ONE launch -- the rows are normalised as they are loaded (same bits as
            # the GroupNorm-apply pass it replaces, the normalised tensor is never written)
            pq = ps.blocks[0].attn1.qkv
            h, qkv0 = ops.stin_block(x_in, ps.proj_in.w, ps.proj_in.b, pq.wf, pq.bf, eps=pq.eps,
                                     gn=(gs_in[2], gs_in[3], act.HW, ps.norm.g, ps.norm.b, ps.norm.eps))
            ws = False
        elif (STIN_GN and gs_in is not None and gs_in[2] is not None and gn_fuse_ok(x_in) and fold_ok(x_in) and ps.proj_in.b is not None
              and ps.proj_in.w.shape == (x_in.shape[1], x_in.shape[1]) and ps.blocks and ps.blocks[0].view_num is None
              and act.HW % ops.ROWLIN_ROWS == 0 and ops.rowlin_ok(x_in.shape[0], x_in.shape[1], x_in.shape[1], "in")):
            # level 1: GroupNorm + proj_in in one row-resident launch (same bits as gn_apply + proj_in)
            h = ops.rowlin(x_in, ps.proj_in.w, ps.proj_in.b, ln=False, gn=(gs_in[2], gs_in[3], act.HW, ps.norm.g, ps.norm.b, ps.norm.eps))
            ws = False
        else:
            if forking(x_in):
                h, xa = gn_fork(Act(x_in, act.N, act.H, act.W, gs=gs_in), ps.norm, False)
                h, x_in = h.tok, xa.tok
            else:
                h = gn(Act(x_in, act.N, act.H, act.W, gs=gs_in), ps.norm, False).tok
            if stin_fused(h, ps):
                # one launch: proj_in + LayerNorm + the block's fused q|k|v projection (level 0); x1 and qkv leave while the next
                # columns multiply
                pq = ps.blocks[0].attn1.qkv
                h, qkv0 = ops.stin_block(h, ps.proj_in.w, ps.proj_in.b, pq.wf, pq.bf, eps=pq.eps)
                ws = False
            elif (fold_ok(h) and ps.proj_in.b is not None and ps.proj_in.w.shape == (h.shape[1], h.shape[1]) and ps.blocks
                  and ps.blocks[0].view_num is None and ops.rowlin_ok(h.shape[0], h.shape[1], h.shape[1], "in")):
                # proj_in with the rows resident in registers (levels 1 / 2); its consumers normalise their rows themselves or take the
                # LayerNorm kernel (multi-view forms), so no row statistics ride on it
                h = ops.rowlin(h, ps.proj_in.w, ps.proj_in.b, ln=False)
                ws = False
            else:
                ws = fold_ok(h)
                h = linear(h, ps.proj_in, want_stats=ws)
    h, st = h if ws else (h, None)
    if dup:
        assert st_dup_ok(ps)
        x_in = dup2(x_in)
        act = Act(x_in, 2 * act.N, act.H, act.W)
    want = gn_fuse_ok(h)      # inference: proj_out rides behind the last feed-forward (fused block / composed GEMM) with GroupNorm sums
    for i, pt in enumerate(ps.blocks):
        kv = kv_cache[pt.kv_slot] if kv_cache is not None else None
        last = i + 1 == len(ps.blocks)
        post = None
        # level 0 with the split feed-forward (row-resident GEGLU projection + one composed GEMM) instead of the fused block
        split = (last and FF_PROJ and want and pt.view_num is None and fold_ok(h) and ps._ff_src is not None
                 and ops.rowlin_ok(act.N * act.HW, h.shape[1], pt.geglu_wf.shape[0], "geglu") and h.shape[1] == ops.FFN_C)
        if last and ps.proj_out_x is not None and not split:
            post = (ps.proj_out_x, ps.proj_out.b, x_in, want and act.HW % ops.FFN_ROWS == 0, act.HW)
        elif last and FF_PROJ and want and ps.ff_proj_w is not None:
            post = ("compose", ps.ff_proj_w, ps.ff_proj_b, x_in, act.HW)      # proj_out composed with the last feed-forward Linear
        r = transformer_block(h, ctx, pt, act.N, act.HW, Lc, kv, st=st, want_stats=not last, dup=dup and i == 0, post=post,
                              qkv0=qkv0 if i == 0 else None, ffn_split=split)
        if isinstance(r[0], str):           # "post": proj_out + x_in ran behind the block's feed-forward
            return Act(r[1], act.N, act.H, act.W, gs=r[2])
        h, st = r
    want = gn_stats_ok(h)
    y = ops.gemm_conv(h, ps.proj_out.w, B=1, H=1, W=h.shape[0], taps=1, bias=ps.proj_out.b, resid=x_in, want_gn_stats=want,
                      gn_hw=act.HW)
    y, gs = y if want else (y, None)
    return Act(y, act.N, act.H, act.W, gs=gs)


def to_tokens(x):
    """[B, L, C] any float dtype -> ([B*L, C] fp16 contiguous, B, L)"""
    B, L, C = x.shape
    return x.reshape(B * L, C).to(compute_dtype()).contiguous(), B, L


def act_from_nchw(x, cpad=None):
    N, C, H, W = x.shape
    return Act(ops.nchw_to_nhwc(x, cpad=cpad, dtype=compute_dtype()), N, H, W)


def act_to_nchw(act: Act, C=None, dtype=None):
    tok = act.materialize()
    return ops.nhwc_to_nchw(tok, act.N, act.H, act.W, C or tok.shape[1], dtype)
