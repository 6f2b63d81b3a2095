"""`ldm.modules.attention` on the MI355X kernels.

Same class names, constructor arguments, forward signatures and state-dict keys as the reference
(ldm/modules/attention.py): GEGLU 51-58, FeedForward 61-78, Normalize 90-91, CrossAttention 147-196,
MemoryEfficientCrossAttention 199-250, BasicTransformerBlock 253-283, SpatialTransformer 331-419.

The nn.Linear / nn.LayerNorm / nn.GroupNorm children are parameter containers only: `forward` re-lays the weights
out once (leftrefill_amd.engine.Packed*) and runs fused HIP kernels on token-major fp16 activations:
  LayerNorm -> fused QKV GEMM (MFMA) -> flash attention -> out-proj GEMM (+bias +residual in the epilogue)
  LayerNorm -> Q GEMM, context KV GEMM -> flash attention (77 keys) -> out-proj GEMM (+residual)
  LayerNorm -> GEGLU GEMM (gate fused in the epilogue) -> FF GEMM (+residual)
There is no eager/PyTorch fallback: without the HIP library the call raises.
"""
import torch
from torch import nn

from leftrefill_amd import engine, ops
from ldm.modules.diffusionmodules.util import checkpoint, zero_module  # noqa: F401  (re-exported like the reference)

XFORMERS_IS_AVAILBLE = False  # the fused HIP attention replaces both reference code paths


def exists(val):
    return val is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def _cached(module, key, build):
    """Pack-once cache, invalidated when any parameter of `module` is replaced or modified in place."""
    sig = tuple((p.data_ptr(), p._version) for p in module.parameters())
    c = module.__dict__.get("_lr_cache")
    if c is None or c[0] != sig:
        c = (sig, {})
        module.__dict__["_lr_cache"] = c
    if key not in c[1]:
        c[1][key] = build()
    return c[1][key]


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        from leftrefill_amd import packing
        tok, B, L = engine.to_tokens(x)
        w, b = _cached(self, "geglu", lambda: packing.pack_geglu(self.proj.weight.detach(), self.proj.bias.detach()))
        y = ops.gemm_conv(tok, w, B=1, H=1, W=tok.shape[0], taps=1, bias=b, geglu=True)
        return y.reshape(B, L, -1).to(x.dtype)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        if not glu:
            raise NotImplementedError("LeftRefill uses the gated (GEGLU) feed-forward only")
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))

    def forward(self, x):
        h = self.net[0](x)
        tok, B, L = engine.to_tokens(h)
        pl = _cached(self, "ff2", lambda: engine.PackedLinear(self.net[2]))
        return engine.linear(tok, pl).reshape(B, L, -1).to(x.dtype)


class CrossAttention(nn.Module):
    """softmax(q k^T / sqrt(d)) v with to_q/to_k/to_v (no bias) and to_out[0] (bias); d_head must be 64."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    def forward(self, x, context=None, mask=None, **kwargs):
        if exists(mask):
            raise NotImplementedError("attention mask is unused by LeftRefill (attention.py:183-187)")
        tok, B, L = engine.to_tokens(x)
        if context is None:
            pa = _cached(self, "self", lambda: engine.PackedAttn(self, True))
            y = engine.attention_plain(tok, None, pa, B, L)
        else:
            ctok, Bc, Lc = engine.to_tokens(context)
            assert Bc == B
            pa = _cached(self, "cross", lambda: engine.PackedAttn(self, False))
            y = engine.attention_plain(tok, ctok, pa, B, L, Lc)
        return y.reshape(B, L, -1).to(x.dtype)


class MemoryEfficientCrossAttention(CrossAttention):
    """Name kept for `ldm.modules.diffusionmodules.model` (reference model.py:10); same fused kernel."""


class BasicTransformerBlock(nn.Module):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                                    context_dim=context_dim if disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def _packed(self):
        return _cached(self, "tblock", lambda: engine.PackedTBlock(self))

    def forward(self, x, context=None):
        tok, B, L = engine.to_tokens(x)
        ctok, _, Lc = engine.to_tokens(context)
        y, _ = engine.transformer_block(tok, ctok, self._packed(), B, L, Lc)
        return y.reshape(B, L, -1).to(x.dtype)


class SpatialTransformer(nn.Module):
    """GroupNorm(eps 1e-6) -> proj_in -> transformer blocks -> proj_out -> + x_in, on NHWC tokens (no transposes)."""

    block_cls = BasicTransformerBlock

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True, one_attn=False, num_patches=None, **block_kwargs):
        super().__init__()
        if one_attn:
            raise NotImplementedError("one_attn is dead code in LeftRefill (attention.py:286)")
        if exists(context_dim) and not isinstance(context_dim, list):
            context_dim = [context_dim]
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        if inner_dim != in_channels:
            raise NotImplementedError("inner_dim != in_channels is not used by LeftRefill configs")
        self.norm = Normalize(in_channels)
        self.one_attn = one_attn
        if use_linear:
            self.proj_in = nn.Linear(in_channels, inner_dim)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            self.block_cls(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                           disable_self_attn=disable_self_attn, checkpoint=use_checkpoint, **block_kwargs)
            for d in range(depth)])
        if use_linear:
            self.proj_out = zero_module(nn.Linear(in_channels, inner_dim))
        else:
            self.proj_out = zero_module(nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        self.use_linear = use_linear

    def _packed(self):
        return _cached(self, "st", lambda: engine.PackedST(self))

    def _fwd(self, act, ctx_tok, Lc):
        return engine.spatial_transformer(act, ctx_tok, Lc, self._packed())

    def forward(self, x, context=None, **kwargs):
        if isinstance(context, list):
            context = context[0]
        act = engine.act_from_nchw(x)
        ctok, _, Lc = engine.to_tokens(context)
        out = self._fwd(act, ctok, Lc)
        return engine.act_to_nchw(out, dtype=torch.float16).to(x.dtype)
