"""Torch-tensor front end of the C ABI: shape/dtype checks, output allocation, current-stream plumbing.

PyTorch is used only for device memory and streams (`torch.cuda.current_stream()` is the HIP stream on ROCm).
Activations are NHWC / token-major fp16: a tensor [M, C] with M = N*H*W rows.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import FfnArgs, GemmArgs, RowlinArgs, StinArgs, XattnArgs

GN_CHUNKS = 256
# per-group GroupNorm partials from the producers' epilogues (no lr_groupnorm_finalize launch); LEFTREFILL_GN_GROUPS=0 keeps the
# per-channel partials + finalize for every GroupNorm
GN_GROUPS = os.environ.get("LEFTREFILL_GN_GROUPS", "1") != "0"

# ---- tile plan of lr_gemm_conv_f16 ----------------------------------------------------------------------------------
# The UNet has ~60 distinct static GEMM shapes.  Their (tile_m, tile_n, splits) come from an IN-TREE table
# (tile_table.json, produced on an MI355X by tools/tune_tiles.py and committed): the plan is a pure function of the
# shape, identical in every process and on every rank, so results are bit-reproducible across runs (the split-K factor
# changes fp32 rounding) and the replicated rows of the sharded multi-view path are bit-identical on every rank.
# Shapes missing from the table use the library's static heuristic (also a pure function of the shape).
# LEFTREFILL_AUTOTUNE=1 is a developer mode: unknown shapes are timed on first sight and added to the in-memory table
# (dump it with tile_cache()); it is never on by default.
AUTOTUNE = os.environ.get("LEFTREFILL_AUTOTUNE", "0") == "1"
# (tile_m, tile_n, pipe): pipe 0 = the tile's standard instance; 4 = the 8-wave 4-stage one-block-per-CU 128-row instance
# (lr_gemm_args.pipe)
# 8 = LR_PIPE_HALO: the halo-tile 3x3 conv (16 x 16 pixel tiles, input patch resident in LDS; refused for every other kind of call)
TILE_CANDIDATES = ((128, 64, 0), (128, 128, 0), (128, 160, 0), (128, 128, 4), (128, 160, 4), (256, 128, 0), (256, 160, 0),
                   (256, 256, 0), (256, 320, 0), (256, 160, 8), (256, 320, 8))
TILE_TABLE_PATH = os.environ.get("LEFTREFILL_TILE_TABLE_PATH",      # (developer override: A/B of a freshly tuned table)
                                 os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_table.json"))
_tile_cache = None
PLAN_LOG = os.environ.get("LEFTREFILL_PLAN_LOG", "0") == "1"
# lr_gemm_args.splitk_mode: 1 = split-K partials reduced inside the GEMM launch where the plan allows it (every K-slice block resident
# at once), 0 = by the separate fixed-order reduce launch
SPLITK_MODE = int(os.environ.get("LEFTREFILL_SPLITK_MODE", "0"))      # (1 needs a developer build of the library)
_untabulated = set()
# shape keys of the GEMM calls since the sets were last cleared, by whether the in-tree table knew them (bench.py batch_sensitivity)
TABLE_HITS, TABLE_MISSES = set(), set()
# developer hooks of tools/tune_in_step.py (None in the product): PLAN_TRIAL maps a table key to the (tile_m, tile_n, splits, pipe) to try
# instead of the table's plan (splits 0 = the library's static choice for that tile; a plan the library refuses falls back to the table's);
# LAUNCH_HOOK(key, phase, plan) is called right before (0) and after (1) the launch
PLAN_TRIAL = None
LAUNCH_HOOK = None


# Plan GEMMs as if the batch were `scale` times larger (the shared prefix of a CFG batch runs on half the samples but must
# form every partial sum exactly like the full batch would: same tile, same split-K, same statistics partials).
_PLAN_BATCH_SCALE = [1]


class plan_batch_scale:
    def __init__(self, scale):
        self.scale = int(scale)

    def __enter__(self):
        self.prev = _PLAN_BATCH_SCALE[0]
        _PLAN_BATCH_SCALE[0] = self.scale

    def __exit__(self, *exc):
        _PLAN_BATCH_SCALE[0] = self.prev


def tile_key(M, N, K, taps=1, stride=1, up=0, geglu=False, concat=False, asym=False, gelu=False, ln=False, stats=False):
    return "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d" % (M, N, K, taps, stride, up, int(geglu), int(concat), int(asym),
                                                      int(gelu), int(ln), int(stats))


def tile_cache():
    """shape key -> [tile_m, tile_n, splits]; loaded once from the in-tree table."""
    global _tile_cache
    if _tile_cache is None:
        _tile_cache = {}
        if os.path.exists(TILE_TABLE_PATH) and os.environ.get("LEFTREFILL_TILE_TABLE", "1") != "0":
            import json
            with open(TILE_TABLE_PATH) as f:
                _tile_cache = {k: tuple(v) for k, v in json.load(f).items()}
    return _tile_cache


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


HALF_TYPES = (torch.float16, torch.bfloat16)


def _chk16(t, name):
    assert t.is_cuda and t.dtype in HALF_TYPES and t.is_contiguous(), f"{name}: need contiguous cuda fp16 / bf16"


def _fn(lib, name, dtype):
    return _lib.fn(lib, name, dtype)


def nchw_to_nhwc(x1, x2=None, cpad=None, dtype=torch.float16):
    """[N,C1,H,W] fp32 (+ optional [N,C2,H,W]) -> [N*H*W, cpad] fp16 (or bf16)."""
    lib = _lib.load()
    N, C1, H, W = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    cpad = cpad or ((C1 + C2 + 7) // 8) * 8
    x1 = x1.float().contiguous()
    x2 = None if x2 is None else x2.float().contiguous()
    y = torch.empty(N * H * W, cpad, device=x1.device, dtype=dtype)
    _lib.check(_fn(lib, "lr_nchw_f32_to_nhwc_f16", dtype)(_p(x1), C1, _p(x2), C2, _p(y), cpad, N, H, W, _stream()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(y, N, H, W, C, out_dtype=None):
    """[N*H*W, >=C] -> [N,C,H,W] in out_dtype: float32, or the 16-bit type of y (default)."""
    lib = _lib.load()
    _chk16(y, "y")
    out_dtype = y.dtype if out_dtype is None or out_dtype in HALF_TYPES else out_dtype
    out = torch.empty(N, C, H, W, device=y.device, dtype=out_dtype)
    _lib.check(_fn(lib, "lr_nhwc_f16_to_nchw", y.dtype)(_p(y), y.shape[-1], C, _p(out), int(out_dtype == torch.float32), N, H, W,
                                       _stream()), "nhwc_to_nchw")
    return out


def group_norm(x1, N, HW, gamma, beta, eps, silu, x2=None):
    """GroupNorm(32)(+SiLU) over the virtual concat [x1 | x2]; x* are [N*HW, C*] fp16 -> [N*HW, C1+C2] fp16."""
    lib = _lib.load()
    _chk16(x1, "x1")
    C1 = x1.shape[-1]
    C2 = 0
    if x2 is not None:
        _chk16(x2, "x2")
        C2 = x2.shape[-1]
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C1 + C2
    partials = torch.empty(N * GN_CHUNKS * 64, device=x1.device, dtype=torch.float32)
    y = torch.empty(N * HW, C1 + C2, device=x1.device, dtype=x1.dtype)
    st = _stream()
    _lib.check(_fn(lib, "lr_groupnorm_stats", x1.dtype)(_p(x1), C1, _p(x2), C2, N, HW, _p(partials), st), "groupnorm_stats")
    _lib.check(_fn(lib, "lr_groupnorm_apply", x1.dtype)(_p(x1), C1, _p(x2), C2, N, HW, _p(partials), _p(gamma), _p(beta), float(eps),
                                      int(bool(silu)), _p(y), st), "groupnorm_apply")
    return y


def group_norm_fused(x1, N, HW, gamma, beta, eps, silu, gs1, x2=None, gs2=None):
    """GroupNorm(32)(+SiLU) whose statistics come from the producers' epilogues: gs* = (partials [M/R, C, 2] fp32, R) as
    returned by gemm_conv(..., want_gn_stats=True) for x1 / x2.  One tiny finalize launch replaces the pass over x."""
    lib = _lib.load()
    _chk16(x1, "x1")
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    p1, r1 = gs1[:2]
    p2, r2 = gs2[:2] if x2 is not None else (None, 1)
    assert p1.shape == (N * HW // r1, C1, 2) and HW % r1 == 0
    if x2 is not None:
        _chk16(x2, "x2")
        assert p2.shape == (N * HW // r2, C2, 2) and HW % r2 == 0
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C1 + C2
    partials = torch.empty(N * 64, device=x1.device, dtype=torch.float32)
    y = torch.empty(N * HW, C1 + C2, device=x1.device, dtype=x1.dtype)
    st = _stream()
    _lib.check(lib.lr_groupnorm_finalize(_p(p1), C1, r1, _p(p2), C2, r2, N, HW, _p(partials), st), "groupnorm_finalize")
    _lib.check(_fn(lib, "lr_groupnorm_apply_n", x1.dtype)(_p(x1), C1, _p(x2), C2, N, HW, _p(partials), 1, _p(gamma), _p(beta), float(eps),
                                        int(bool(silu)), _p(y), st), "groupnorm_apply_n")
    return y


def group_norm_groups(x1, N, HW, gamma, beta, eps, silu, gp, chunks):
    """GroupNorm(32)(+SiLU) of a single tensor whose per-GROUP (sum, sumsq) partials gp [N, chunks, 32, 2] came out of its producer's
    epilogue (gemm_conv / ffn_block `want_gn_stats`): ONE launch, the apply kernel reduces the chunks in its prologue."""
    lib = _lib.load()
    _chk16(x1, "x1")
    C1 = x1.shape[-1]
    assert gp.dtype == torch.float32 and gp.is_contiguous() and gp.shape == (N, chunks, 32, 2), (gp.shape, N, chunks)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C1
    y = torch.empty(N * HW, C1, device=x1.device, dtype=x1.dtype)
    _lib.check(_fn(lib, "lr_groupnorm_apply_n", x1.dtype)(_p(x1), C1, 0, 0, N, HW, _p(gp), chunks, _p(gamma), _p(beta), float(eps),
                                                          int(bool(silu)), _p(y), _stream()), "groupnorm_apply_n")
    return y


OUT_FUSED = os.environ.get("LEFTREFILL_OUT_FUSED", "1") != "0"   # the UNet's `out` block as one launch (lr_gn_conv_out_f16)


def gn_conv_out_ok(H, W, C, cout):
    """Shapes lr_gn_conv_out_f16 takes (anything else keeps GroupNorm -> conv -> layout conversion)."""
    return OUT_FUSED and 1 <= cout <= 4 and C % 64 == 0 and 64 <= C <= 704 and H % 8 == 0 and W % 16 == 0


def gn_conv_out(x, N, H, W, gamma, beta, eps, gp, chunks, w, bias, cout):
    """GroupNorm(32) + SiLU + 3x3 pad-1 conv to cout <= 4 channels + NHWC -> NCHW in one launch (the UNet's `out` block,
    reference openaimodel.py:714-718).  x [N*H*W, C] 16-bit tokens, gp [N, chunks, 32, 2] group partials of x from its producer,
    w [>= cout, 9*C] packed conv weight (k = tap * C + channel), bias fp32 | None -> [N, cout, H, W] in x's dtype."""
    lib = _lib.load()
    _chk16(x, "x")
    _chk16(w, "w")
    C = x.shape[1]
    assert x.shape[0] == N * H * W and gn_conv_out_ok(H, W, C, cout), (x.shape, N, H, W, cout)
    assert gp.dtype == torch.float32 and gp.is_contiguous() and gp.shape == (N, chunks, 32, 2), (gp.shape, N, chunks)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    assert w.dtype == x.dtype and w.shape[0] >= cout and w.shape[1] == 9 * C and (bias is None or (bias.dtype == torch.float32 and bias.numel() >= cout))
    y = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
    _lib.check(_fn(lib, "lr_gn_conv_out_f16", x.dtype)(_p(x), N, H, W, C, _p(gp), chunks, _p(gamma), _p(beta), float(eps), _p(w), w.stride(0),
                                                       _p(bias), cout, _p(y), _stream()), "gn_conv_out")
    return y


def gn_fold_weights(gp, chunks, N, HW, gamma, beta, eps, w, bias):
    """Per-sample copies of a pointwise layer's weights with GroupNorm(32, affine) folded in (lr_gn_fold_weights_f16):
    w [Nout, C] 16-bit, bias [Nout] fp32 | None -> (w_b [N, Nout, C], bias_b [N, Nout] fp32); gp / chunks as in group_norm_groups."""
    lib = _lib.load()
    if not _lib.dev_variants():
        raise RuntimeError("lr_gn_fold_weights_f16 is compiled in developer builds only (tools/build_variant.sh)")
    _chk16(w, "w")
    Nout, C = w.shape
    assert gp.dtype == torch.float32 and gp.is_contiguous() and gp.shape == (N, chunks, 32, 2)
    wb = torch.empty(N, Nout, C, device=w.device, dtype=w.dtype)
    bb = torch.empty(N, Nout, device=w.device, dtype=torch.float32)
    _lib.check(_fn(lib, "lr_gn_fold_weights_f16", w.dtype)(_p(gp), chunks, N, HW, C, _p(gamma), _p(beta), float(eps), _p(w), _p(bias), Nout,
                                                           _p(wb), _p(bb), _stream()), "gn_fold_weights")
    return wb, bb


def layer_norm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    _chk16(x, "x")
    M, C = x.shape
    y = torch.empty_like(x)
    _lib.check(_fn(lib, "lr_layernorm", x.dtype)(_p(x), _p(gamma), _p(beta), float(eps), _p(y), M, C, _stream()), "layernorm")
    return y


def timestep_embedding(t, dim, dtype=torch.float16):
    lib = _lib.load()
    t = t.to(torch.int64).contiguous()
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=dtype)
    _lib.check(_fn(lib, "lr_timestep_embedding", dtype)(_p(t), t.shape[0], dim, _p(out), _stream()), "timestep_embedding")
    return out


def linear_small_m(a, w, bias, act_in=False, act_out=False):
    """a [M<=16, K] fp16, w [N, K] fp16, bias [N] fp32 or None -> [M, N] fp16."""
    lib = _lib.load()
    _chk16(a, "a")
    _chk16(w, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    st = _stream()
    for m0 in range(0, M, 16):   # the kernel keeps <= 16 rows in LDS; larger batches go in row chunks
        mc = min(16, M - m0)
        _lib.check(_fn(lib, "lr_linear_small_m", a.dtype)(a[m0:].data_ptr(), K, _p(w), _p(bias), out[m0:].data_ptr(), N, mc, N, K,
                                         int(act_in), int(act_out), st), "linear_small_m")
    return out


def gemm_conv(x1, wt, *, B, H, W, Hs=None, Ws=None, taps=1, stride=1, up=0, asym=False, x2=None, bias=None, rowvec=None,
              resid=None, geglu=False, gelu=False, out=None, tile_n=0, tile_m=0, splits=0, pipe=0, ln=None,
              want_stats=False, want_gn_stats=False, gn_hw=0, per_sample=False, wt_pm=False, skip=None):
    """Implicit-GEMM conv / linear (see lr_gemm_conv_f16).  x1 [B*Hs*Ws, C1] fp16, wt [N, taps*(C1+C2)] fp16.

    skip = (s1, s2 | None): pointwise K extension over the virtual concat [s1 | s2] at the output resolution -- wt is
    [N, 9 (C1 + C2) + Cs1 + Cs2], out = conv3x3([x1 | x2]) + W_s [s1 | s2] in one accumulation (lr_gemm_args.skip1: a ResBlock's
    skip_connection inside its last conv); bias must be the sum of the two layers' biases.

    ln = (stats [M, parts, 2] fp32, eps, colsum [N] fp32): LayerNorm folded into the GEMM (x1 is the raw input, wt / bias
    are the gamma / beta folded weights, see lr_gemm_args).  want_stats: also return the per-row (sum, sumsq) partials of
    the output, [M, parts, 2] fp32 -- the `stats` a following LayerNorm-folded GEMM consumes; returns (out, stats).
    want_gn_stats: also return per-channel (sum, sumsq) over row blocks for the GroupNorm that consumes the output:
    returns (out, (partials [M / R, N, 2] fp32, R, gp, chunks)), or (out, None) when R does not divide the rows of a sample
    (gn_hw, default H*W); gp [samples, chunks, 32, 2] = per-GROUP sums for a consumer that normalises this tensor alone
    (group_norm_groups: no finalize launch), None when the plan cannot produce them.
    per_sample: wt is [B, N, K] and bias [B, N] -- one weight set per sample (rows b*H*W .. of sample b), see ops.gn_fold_weights."""
    lib = _lib.load()
    _chk16(x1, "x1")
    _chk16(wt, "wt")
    Hs = H if Hs is None else Hs
    Ws = W if Ws is None else Ws
    C1 = x1.shape[-1]
    C2 = 0
    if x2 is not None:
        _chk16(x2, "x2")
        C2 = x2.shape[-1]
    if per_sample:
        assert wt.dim() == 3 and wt.shape[0] == B and taps == 1 and x2 is None and wt.is_contiguous() and not wt_pm
    if wt_pm:      # piece-major weights [K / 64, N, 64] (packing.pack_pm)
        assert wt.dim() == 3 and wt.shape[2] == 64 and wt.is_contiguous()
        Nw, Kw = wt.shape[1], wt.shape[0] * 64
    else:
        Nw, Kw = wt.shape[-2], wt.shape[-1]
    c16 = C1 == 16 and C2 == 0 and taps == 9 and Kw == 192      # 16-channel 3x3 source: K = 144 zero-padded to three K-steps
    Cs1 = Cs2 = 0
    if skip is not None:
        s1, s2 = skip
        _chk16(s1, "skip1")
        Cs1 = s1.shape[-1]
        if s2 is not None:
            _chk16(s2, "skip2")
            Cs2 = s2.shape[-1]
        assert stride == 1 and not up and s1.shape[0] == B * H * W and (s2 is None or s2.shape[0] == s1.shape[0])
    assert c16 or Kw == taps * (C1 + C2) + Cs1 + Cs2, (wt.shape, taps, C1, C2, Cs1, Cs2)
    if c16 and tile_m == 0 and tile_n == 0:
        # only the pipelined 256-row instances gather 16-channel taps
        tile_m, tile_n, splits = 256, (320 if Nw % 320 == 0 else 160 if Nw % 160 == 0 else 128), 1
    assert x1.shape[0] == B * Hs * Ws, (x1.shape, B, Hs, Ws)
    M = B * H * W
    n_out = Nw // 2 if geglu else Nw
    if out is None:
        out = torch.empty(M, n_out, device=x1.device, dtype=x1.dtype)
    a = GemmArgs()
    a.p1, a.C1, a.p2, a.C2 = _p(x1), C1, _p(x2), C2
    a.B, a.H, a.W, a.Hs, a.Ws = B, H, W, Hs, Ws
    a.taps, a.stride, a.up, a.asym = taps, stride, up, int(bool(asym))
    a.wt, a.N = _p(wt), Nw
    a.bias = _p(bias)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == (Nw * B if per_sample else Nw)
    a.rowvec, a.ld_rowvec = _p(rowvec), (rowvec.stride(0) if rowvec is not None else 0)
    a.resid, a.ld_resid = _p(resid), (resid.stride(0) if resid is not None else 0)
    if resid is not None:
        assert resid.shape[0] == M and resid.dtype == x1.dtype
    a.out, a.ld_out = _p(out), out.stride(0)
    assert not (geglu and gelu)
    a.geglu = 2 if gelu else int(geglu)      # 1: fused GEGLU, 2: plain erf-GELU epilogue
    a.tile_n = tile_n
    a.tile_m = tile_m
    a.splits = splits
    a.pipe = pipe
    a.workspace, a.workspace_bytes = 0, 0
    a.ln_stats, a.ln_parts, a.ln_eps, a.ln_colsum, a.stats_out, a.gn_stats_out = 0, 0, 0.0, 0, 0, 0
    a.gn_group_out, a.gn_hw = 0, int(gn_hw)
    a.wt_bstride, a.bias_bstride = (Nw * Kw, Nw if bias is not None else 0) if per_sample else (0, 0)
    a.wt_pm = int(bool(wt_pm))
    a.skip1, a.skip2, a.Cs1, a.Cs2 = (_p(skip[0]), _p(skip[1]), Cs1, Cs2) if skip is not None else (0, 0, 0, 0)
    a.splitk_mode = SPLITK_MODE
    assert wt.dtype == x1.dtype, (wt.dtype, x1.dtype)
    a.dtype = int(x1.dtype == torch.bfloat16)      # LR_DTYPE_F16 | LR_DTYPE_BF16
    if ln is not None:
        st_in, eps, colsum = ln
        assert st_in.dtype == torch.float32 and st_in.is_contiguous() and st_in.shape[0] == M and st_in.shape[2] == 2
        assert colsum.dtype == torch.float32 and colsum.numel() == Nw and taps == 1 and x2 is None
        a.ln_stats, a.ln_parts, a.ln_eps, a.ln_colsum = _p(st_in), st_in.shape[1], float(eps), _p(colsum)
    st = _stream()
    if tile_m == 0 and tile_n == 0 and splits == 0:
        scale = _PLAN_BATCH_SCALE[0]
        # (a conv with the pointwise extension is planned like the plain conv: the table knows that shape)
        key = tile_key(M * scale, Nw, Kw - Cs1 - Cs2, taps, stride, up, geglu, C2 > 0, asym, gelu, ln is not None, want_stats)
        best = tile_cache().get(key)
        (TABLE_HITS if best is not None else TABLE_MISSES).add(key)
        if best is None and PLAN_LOG and key not in _untabulated:      # developer aid: which shapes fall back to the static heuristic
            _untabulated.add(key)
            import sys
            print(f"[leftrefill] untabulated GEMM shape {key} (taps {taps}, H {H}, W {W}, skip {skip is not None})", file=sys.stderr, flush=True)
        if best is None and (scale != 1 or (HEURISTIC_REFINE and not AUTOTUNE and skip is None and not per_sample and not c16)):
            # not in the table: the library's static heuristic for the (scaled) batch, refined by what the in-step tuning of round 6 found on
            # every small-M shape (_refine_plan); a caller of the C ABI without this front end gets the unrefined heuristic
            a.B = B * scale
            if ln is not None or want_stats:
                a.splits = 1                 # row statistics / the LayerNorm fold never split K -- like the unscaled plan below
            best = _refine_plan(lib, a, geglu or gelu) if HEURISTIC_REFINE and skip is None and not per_sample and not c16 else _lib_plan(lib, a)
            a.B, a.splits, a.tile_m, a.tile_n, a.pipe = B, 0, 0, 0, 0
        if best is None and AUTOTUNE and not torch.cuda.is_current_stream_capturing():
            best = _tune_tiles(lib, a, x1.device, geglu, ln is not None or want_stats, want_stats)
            tile_cache()[key] = best
        if best is not None and PLAN_TRIAL is not None and PLAN_TRIAL.get(key) is not None:
            best = PLAN_TRIAL[key]
        if best is not None:
            a.tile_m, a.tile_n, a.splits = best[:3]
            a.pipe = best[3] if len(best) > 3 else 0
            if a.pipe == 8 and (W % 16 or (H % 16 and not (H == 8 and M % 256 == 0 and (M * scale) % 256 == 0))):      # the table is keyed by M: the halo-tile instance needs 16 x 16 pixel tiles (or pairs of 8-line images; decided for the planned batch AND this call),
                a.pipe = 0                              # any other latent of the same size takes the tile's gather instance
        if skip is not None and best is None:      # static heuristic: ask the library, then make sure the tile is a pipelined one
            plan = (ctypes.c_int32 * 4)()
            lib.lr_gemm_plan(a, plan)
            a.tile_m, a.tile_n, a.splits, a.pipe = plan[0], plan[1], plan[2], plan[3]
    stats = None
    if want_stats:
        if a.splits == 0:
            a.splits = 1     # row statistics come out of the epilogue: no split-K
        parts = lib.lr_gemm_stats_parts(a)
        stats = torch.empty(M, parts, 2, device=x1.device, dtype=torch.float32)
        a.stats_out = _p(stats)
    if ln is not None and a.splits == 0:
        a.splits = 1
    gstats = None
    if want_gn_stats:
        plan = (ctypes.c_int32 * 4)()
        lib.lr_gemm_plan(a, plan)
        a.splits = plan[2]                  # pinned: the row-block size of the statistics depends on it (32 behind split-K)
        rows = lib.lr_gemm_gn_rows(a)
        hw = gn_hw or H * W
        if hw % rows == 0:                  # row blocks never straddle two samples
            chunks = lib.lr_gemm_gn_group_chunks(a) if GN_GROUPS else 0
            gp = torch.empty(M // hw, chunks, 32, 2, device=x1.device, dtype=torch.float32) if chunks > 0 else None
            gstats = (torch.empty((M + rows - 1) // rows, n_out, 2, device=x1.device, dtype=torch.float32), rows, gp, chunks)
            a.gn_stats_out = _p(gstats[0])
            a.gn_group_out = _p(gp)
    ws = _workspace(lib, a, x1.device)
    if LAUNCH_HOOK is not None and tile_m == 0 and tile_n == 0 and splits == 0:
        plan = (ctypes.c_int32 * 4)()
        lib.lr_gemm_plan(a, plan)
        LAUNCH_HOOK(key, 0, tuple(plan))
        rc = lib.lr_gemm_conv_f16(a, st)
        LAUNCH_HOOK(key, 1, rc)
        if rc != 0 and PLAN_TRIAL is not None and PLAN_TRIAL.get(key) is not None:      # the trial plan is not one this call can take
            PLAN_TRIAL[key] = None
            return gemm_conv(x1, wt, B=B, H=H, W=W, Hs=Hs, Ws=Ws, taps=taps, stride=stride, up=up, asym=asym, x2=x2, bias=bias, rowvec=rowvec,
                             resid=resid, geglu=geglu, gelu=gelu, out=out, ln=ln, want_stats=want_stats, want_gn_stats=want_gn_stats, gn_hw=gn_hw,
                             per_sample=per_sample, wt_pm=wt_pm, skip=skip)
        _lib.check(rc, "gemm_conv")
    else:
        _lib.check(lib.lr_gemm_conv_f16(a, st), "gemm_conv")
    if want_gn_stats:
        return out, gstats
    return (out, stats) if want_stats else out


# untabulated shapes: refine the library's static plan (LEFTREFILL_HEURISTIC_REFINE=0: take it as is)
HEURISTIC_REFINE = os.environ.get("LEFTREFILL_HEURISTIC_REFINE", "1") != "0"


def _lib_plan(lib, a):
    plan = (ctypes.c_int32 * 4)()
    lib.lr_gemm_plan(a, plan)
    return tuple(plan)


def _refine_plan(lib, a, act_epilogue):
    """The static heuristic's plan for `a`, with the two regularities of the in-step tuned table applied to 128-row tiles (round 6: with weights
    streamed from HBM the 2-stage 4-wave kernel loses to the 4-stage ring on every shape whose tiles fit the chip in one round, and 160-column
    tiles beat 128 where they divide N): tile_n 160 if N % 160 == 0, the 4-stage ring (pipe 4) if the tiles number <= 256.  A pure function of
    the shape like the table."""
    base = _lib_plan(lib, a)
    tm, tn = base[0], base[1]
    M = a.B * a.H * a.W
    if (a.taps == 9 and a.stride == 1 and a.up == 0 and not a.asym and not act_epilogue and not a.ln_stats and not a.stats_out and a.W % 16 == 0
            and a.H % 16 == 0 and a.Hs == a.H and a.Ws == a.W and a.N % 160 == 0 and M >= 8192 and (a.C1 + a.C2) % 64 == 0 and M < (1 << 28)):
        # third regularity: the 3x3 stride-1 convs of the large levels run on the halo-tile instance (input patch resident in LDS) -- 320-column
        # tiles where they still fill the chip
        tn = 320 if a.N % 320 == 0 and ((M + 255) // 256) * (a.N // 320) >= 200 else 160
        keep = (a.tile_m, a.tile_n, a.pipe)
        a.tile_m, a.tile_n, a.pipe = 256, tn, 8
        plan = _lib_plan(lib, a)
        a.tile_m, a.tile_n, a.pipe = keep
        return (256, tn, plan[2], 8)
    if tm != 128 or act_epilogue:
        return base
    if a.N % 160 == 0:
        tn = 160
    if tn not in (128, 160) or ((M + 127) // 128) * ((a.N + tn - 1) // tn) > 256:
        return base
    keep = (a.tile_m, a.tile_n, a.pipe)
    a.tile_m, a.tile_n, a.pipe = 128, tn, 4
    plan = _lib_plan(lib, a)                 # split-K factor of the refined tile
    a.tile_m, a.tile_n, a.pipe = keep
    return (128, tn, plan[2], 4)


def gemm_plan(M, N, K, **kw):
    """(tile_m, tile_n, splits, pipe) lr_gemm_conv_f16 uses for a shape: the in-tree table, else the static heuristic."""
    best = tile_cache().get(tile_key(M, N, K, **kw))
    if best is not None:
        return tuple(best) if len(best) > 3 else tuple(best) + (0,)
    lib = _lib.load()
    a = GemmArgs()
    a.B, a.H, a.W, a.N, a.taps, a.C1 = 1, 1, M, N, kw.get("taps", 1), K // kw.get("taps", 1)
    a.geglu = int(kw.get("geglu", False))
    if kw.get("ln") or kw.get("stats"):
        a.splits = 1
    plan = _refine_plan(lib, a, bool(kw.get("geglu") or kw.get("gelu"))) if HEURISTIC_REFINE else _lib_plan(lib, a)
    return tuple(plan)[:3] + (plan[3] if plan[3] in (4, 8) else 0,)


def _workspace(lib, a, device):
    """split-K partials (small-M shapes); the caller owns the workspace."""
    a.workspace, a.workspace_bytes = 0, 0
    need = lib.lr_gemm_workspace_bytes(a)
    if need <= 0:
        return None
    ws = torch.empty(need // 4, device=device, dtype=torch.float32)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    return ws


def _time_launch(lib, a, st, reps, rounds):
    t = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.lr_gemm_conv_f16(a, st)
        e1.record()
        e1.synchronize()
        t = min(t, e0.elapsed_time(e1))
    return t


def _tune_tiles(lib, a, device, geglu, no_split, want_stats, reps=4, rounds=2):
    """Developer mode: time every (tile, split-K) configuration and return the fastest (tile_m, tile_n, splits)."""
    st = _stream()
    best, best_t = None, float("inf")
    plan = (ctypes.c_int32 * 4)()
    for tm, tn, stg in TILE_CANDIDATES:
        if geglu and tn == 160:
            continue
        a.tile_m, a.tile_n, a.splits, a.pipe = tm, tn, 0, stg
        lib.lr_gemm_plan(a, plan)
        cands = {1} if no_split else {1, int(plan[2])}
        for sp in sorted(cands):
            a.splits = sp
            stats = None
            if want_stats:
                parts = lib.lr_gemm_stats_parts(a)
                stats = torch.empty(a.B * a.H * a.W, parts, 2, device=device, dtype=torch.float32)
                a.stats_out = stats.data_ptr()
            ws = _workspace(lib, a, device)
            if sp > 1 and ws is None:
                continue
            if lib.lr_gemm_conv_f16(a, st) != 0:
                continue
            t = _time_launch(lib, a, st, reps, rounds)
            del ws, stats
            if t < best_t:
                best, best_t = (tm, tn, sp, stg), t
    a.stats_out = 0
    return best


VT_MIN_KEYS = int(os.environ.get("LEFTREFILL_VT_MIN_KEYS", "1024"))   # pre-transpose V for key sequences at least this long
# 0 (default): V stays in its natural [key][d] layout (lr_attention_f16 gathers the PV fragments with the LDS transpose read of gfx950):
# no lr_transpose_v_f16 copy in front of long-sequence attention, no cached V^T of the context.  1: the round-1..3 path (pre-transposed,
# key-permuted V^T through lr_attention_vt_f16 from VT_MIN_KEYS keys up).  Measured (profiles/r04_attn_tr.txt): the kernel itself is
# 1-2 % slower on natural V, the copy it saves costs 11-18 us per launch -- 8192^2: 693 vs 681 + 18 us, 2048^2: 115 vs 113 + 13 us.
ATTN_VT = os.environ.get("LEFTREFILL_ATTN_VT", "0") != "0"


def transpose_v(v, B, heads, Nkv, out=None):
    """v [B*Nkv, >=heads*64] (row stride ldv) -> V^T [B, heads*64, pad64(Nkv)] in the attention kernel's key order."""
    lib = _lib.load()
    assert v.is_cuda and v.dtype in HALF_TYPES and v.stride(1) == 1
    ld = ((Nkv + 63) // 64) * 64
    vt = torch.empty(B, heads * 64, ld, device=v.device, dtype=v.dtype) if out is None else out
    assert vt.shape == (B, heads * 64, ld) and vt.is_contiguous()
    _lib.check(_fn(lib, "lr_transpose_v_f16", v.dtype)(_p(v), v.stride(0), _p(vt), ld, B, heads, Nkv, _stream()), "transpose_v")
    return vt


def attention(q, k, v, B, heads, Nq, Nkv, scale, out=None, vt=None):
    """q [B*Nq, >=heads*64] (row stride = ldq), k/v [B*Nkv, ...]; returns [B*Nq, heads*64] fp16.

    q/k/v may be column slices of a fused projection (strided rows, unit column stride).  vt: optional transpose_v(v) -- the
    pre-transposed-V kernel runs when it is given (or, with ATTN_VT, made here for long key sequences)."""
    lib = _lib.load()
    for t_ in (q, k, v):
        assert t_.is_cuda and t_.dtype == q.dtype and q.dtype in HALF_TYPES and t_.stride(1) == 1
    if out is None:
        out = torch.empty(B * Nq, heads * 64, device=q.device, dtype=q.dtype)
    if vt is None and ATTN_VT and Nkv >= VT_MIN_KEYS:
        vt = transpose_v(v, B, heads, Nkv)
    if vt is not None:
        _lib.check(_fn(lib, "lr_attention_vt_f16", q.dtype)(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), vt.shape[2], _p(out),
                                           out.stride(0), B, heads, Nq, Nkv, float(scale), _stream()), "attention_vt")
        return out
    _lib.check(_fn(lib, "lr_attention_f16", q.dtype)(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                    B, heads, Nq, Nkv, float(scale), _stream()), "attention")
    return out


def attention_causal(q, k, v, B, heads, N, scale):
    """Causal self-attention (query i sees keys <= i), q/k/v [B*N, >=heads*64] strided column slices; the text tower."""
    lib = _lib.load()
    for t_ in (q, k, v):
        assert t_.is_cuda and t_.dtype == q.dtype and q.dtype in HALF_TYPES and t_.stride(1) == 1
    out = torch.empty(B * N, heads * 64, device=q.device, dtype=q.dtype)
    _lib.check(_fn(lib, "lr_attention_causal_f16", q.dtype)(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                           B, heads, N, float(scale), _stream()), "attention_causal")
    return out


def attention_qkv(qkv, B, heads, L, scale):
    """Self-attention on the fused projection qkv [B*L, 3C] (q | k | v column blocks)."""
    C = heads * 64
    return attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, heads, L, L, scale)


def attention_q_kv(q, kv, B, heads, Nq, Nkv, scale, vt=None):
    """Cross-attention: q [B*Nq, C], fused context projection kv [B*Nkv, 2C] (k | v column blocks); vt: optional cached
    transpose_v of the v block (the context is constant over the DDIM steps)."""
    C = heads * 64
    return attention(q, kv[:, :C], kv[:, C:], B, heads, Nq, Nkv, scale, vt=vt)


XATTN_C, XATTN_ROWS, XATTN_MAX_KEYS = 320, 128, 96
# widths lr_xattn_block_f16 has an instance for -> rows per block (C = 640: the 64-row / 4-wave instance of xattn_block640.hip)
XATTN_WIDTHS = {320: 128, 640: 64}
# LEFTREFILL_XATTN_WIDE=0 keeps the to_q -> attention -> to_out launches at C = 640
XATTN_WIDE = os.environ.get("LEFTREFILL_XATTN_WIDE", "1") != "0"


def xattn_ok(M, HW, C, heads, Lc):
    """Shapes lr_xattn_block_f16 takes (everything else keeps the to_q -> attention -> to_out path)."""
    rows = XATTN_WIDTHS.get(C)
    if rows is None or (C != XATTN_C and not XATTN_WIDE):
        return False
    return heads * 64 == C and M % rows == 0 and HW % rows == 0 and 0 < Lc <= XATTN_MAX_KEYS


def xattn_pack_vt(v, B, heads, Lc, out=None):
    """v [B*Lc, >= heads*64] (V projection of the context, unit column stride) -> V^T pack [B, heads, 2, 64, 64]."""
    lib = _lib.load()
    assert v.is_cuda and v.dtype in HALF_TYPES and v.stride(1) == 1 and v.shape[0] == B * Lc
    vt = torch.empty(B, heads, 2, 64, 64, device=v.device, dtype=v.dtype) if out is None else out
    assert vt.shape == (B, heads, 2, 64, 64) and vt.is_contiguous()
    _lib.check(_fn(lib, "lr_xattn_pack_vt_f16", v.dtype)(_p(v), v.stride(0), _p(vt), B, heads, Lc, _stream()), "xattn_pack_vt")
    return vt


def xattn_block(x, wq, bq, k, vt, wo, bo, *, HW, heads, Lc, eps, scale, want_stats=False, out=None, pre=None):
    """x + to_out(attention(LayerNorm(x) Wq, K, V)) in one launch (lr_xattn_block_f16).  wq / bq: LayerNorm-folded to_q;
    k [B*Lc, ld] = context projection with packing.pack_xattn's row order; vt = xattn_pack_vt(V); wo: packing.pack_pieces(to_out).
    pre = (a, Wo1, bo1): also fuses the preceding self-attention's out-projection, x1 = a Wo1^T + bo1 + x (wq: xattn_perm columns).
    Returns out [M, C] (, stats [M, 1, 2] for the LayerNorm fold of the next GEMM)."""
    lib = _lib.load()
    _chk16(x, "x")
    M, C = x.shape
    assert xattn_ok(M, HW, C, heads, Lc), (M, HW, C, heads, Lc)
    assert wq.dtype == x.dtype and wq.is_contiguous() and wq.shape == (C, C)
    assert wo.dtype == x.dtype and wo.is_contiguous() and wo.shape == (heads, C, 64)
    assert bq.dtype == torch.float32 and bo.dtype == torch.float32 and bq.numel() == C and bo.numel() == C
    assert k.dtype == x.dtype and k.stride(1) == 1 and k.shape[0] == (M // HW) * Lc and vt.dtype == x.dtype and vt.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous() and out.device == x.device
    stats = torch.empty(M, 1, 2, device=x.device, dtype=torch.float32) if want_stats else None
    a = XattnArgs()
    a.x, a.out, a.wq, a.bq, a.k, a.ldk, a.vt, a.wo, a.bo = _p(x), _p(out), _p(wq), _p(bq), _p(k), k.stride(0), _p(vt), _p(wo), _p(bo)
    a.stats_out = _p(stats)
    a.M, a.HW, a.C, a.heads, a.Lc, a.ln_eps, a.scale = M, HW, C, heads, Lc, float(eps), float(scale)
    a.pre_a = a.pre_w = a.pre_b = 0
    if pre is not None:
        # (a_self, Wo1, bo1): x1 = a_self Wo1^T + bo1 + x runs in front, in the same launch; wq then has pi-ordered COLUMNS
        pa_, pw_, pb_ = pre
        _chk16(pa_, "pre_a")
        assert pa_.shape == (M, C) and pw_.dtype == x.dtype and pw_.is_contiguous() and pw_.shape == (C, C)
        assert pb_.dtype == torch.float32 and pb_.numel() == C
        a.pre_a, a.pre_w, a.pre_b = _p(pa_), _p(pw_), _p(pb_)
    _lib.check(_fn(lib, "lr_xattn_block_f16", x.dtype)(a, _stream()), "xattn_block")
    return (out, stats) if want_stats else out


STIN_C, STIN_ROWS = 320, 256


def stin_ok(M, C, NQ):
    """Shapes lr_stin_block_f16 takes (everything else keeps the proj_in GEMM -> LayerNorm-folded q|k|v GEMM path)."""
    return C == STIN_C and M % STIN_ROWS == 0 and NQ > 0 and NQ % 192 == 0 and NQ <= 4096


def stin_block(x, wp, bp, wqkv, bqkv, *, eps, out=None, qkv_out=None, gn=None):
    """x1 = x wp^T + bp;  qkv = LayerNorm(x1) wqkv^T + bqkv  in one launch (lr_stin_block_f16): SpatialTransformer.proj_in and the
    LayerNorm-folded fused q|k|v projection of its block's self-attention.  wqkv / bqkv: packing.fold_layernorm of [to_q; to_k; to_v]
    (natural column order).  gn = (gp [samples, chunks, 32, 2], chunks, HW, gamma, beta, eps): x is the RAW tensor and the SpatialTransformer's
    GroupNorm is applied to the rows on the way in (what group_norm_groups would have written, bit for bit).  Returns (x1 [M, C], qkv [M, NQ])."""
    lib = _lib.load()
    _chk16(x, "x")
    M, C = x.shape
    NQ = wqkv.shape[0]
    assert stin_ok(M, C, NQ), (M, C, NQ)
    assert wp.dtype == x.dtype and wp.is_contiguous() and wp.shape == (C, C) and wqkv.dtype == x.dtype and wqkv.is_contiguous() and wqkv.shape == (NQ, C)
    assert bp.dtype == torch.float32 and bp.numel() == C and bqkv.dtype == torch.float32 and bqkv.numel() == NQ
    x1 = torch.empty_like(x) if out is None else out
    qkv = torch.empty(M, NQ, device=x.device, dtype=x.dtype) if qkv_out is None else qkv_out
    assert x1.shape == x.shape and x1.is_contiguous() and qkv.shape == (M, NQ) and qkv.stride(1) == 1 and qkv.dtype == x.dtype
    a = StinArgs()
    a.x, a.wp, a.bp, a.wqkv, a.bqkv, a.x1, a.qkv = _p(x), _p(wp), _p(bp), _p(wqkv), _p(bqkv), _p(x1), _p(qkv)
    a.M, a.C, a.NQ, a.ld_qkv, a.ln_eps = M, C, NQ, qkv.stride(0), float(eps)
    a.gn_part = a.gn_gamma = a.gn_beta = 0
    a.gn_chunks = a.gn_hw = 0
    a.gn_eps = 0.0
    if gn is not None:
        gp, chunks, hw, gamma, beta, geps = gn
        assert gp.dtype == torch.float32 and gp.is_contiguous() and gp.shape == (M // hw, chunks, 32, 2) and hw % STIN_ROWS == 0
        assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
        a.gn_part, a.gn_gamma, a.gn_beta, a.gn_chunks, a.gn_hw, a.gn_eps = _p(gp), _p(gamma), _p(beta), int(chunks), int(hw), float(geps)
    _lib.check(_fn(lib, "lr_stin_block_f16", x.dtype)(a, _stream()), "stin_block")
    return x1, qkv


ROWLIN_ROWS = 128
# widths the row-resident kernel takes, by kind of projection: "qkv" (LayerNorm + fused q|k|v), "q" (LayerNorm + attn2.to_q), "in" (plain
# proj_in), "geglu" (LayerNorm + gated projection).  Level 1 only: at level 2 (C = 1280, 4096 rows) it was measured and lost to the tiled
# GEMM (31.9 vs 24.5 us at N = 1280, 58.2 vs 54.8 q|k|v, 128 vs 113 GEGLU; profiles/r06_rowlin_microbench.txt) -- that instance is
# compiled in developer builds only.
ROWLIN_WIDTHS = {"qkv": (640,), "q": (), "in": (640,), "geglu": (640,)}
# level 0 (C = 320), developer builds of the library only: the feed-forward as row-resident GEGLU projection (256-row blocks) + ONE GEMM for
# ff.net[2] composed with proj_out, instead of the fused lr_ffn_block_f16.  Measured and lost: 129.5 + 99.1 us against 220-225 us, UNet step
# 17.92 vs 17.87 ms same box (profiles/r06_ffn_split_ab.txt) -- off; LEFTREFILL_FFN_SPLIT=1 with LEFTREFILL_LIB_PATH=<developer build> enables it.
FFN_SPLIT = os.environ.get("LEFTREFILL_FFN_SPLIT", "0") != "0"
ROWLIN_MIN_ROWS = 2048      # below this the column slices of a launch cannot fill the chip
# LEFTREFILL_ROWLIN=0 keeps the tiled GEMMs everywhere
ROWLIN = os.environ.get("LEFTREFILL_ROWLIN", "1") != "0"


def rowlin_ok(M, C, N, kind="qkv"):
    """Shapes / uses lr_rowlin_f16 takes (everything else keeps the [LayerNorm-folded] GEMM)."""
    if kind == "geglu" and C == 320:
        return ROWLIN and FFN_SPLIT and _lib.dev_variants() and M % (2 * ROWLIN_ROWS) == 0 and M >= 8 * ROWLIN_MIN_ROWS and N > 0 and N % 64 == 0 and N <= 2560
    return ROWLIN and C in ROWLIN_WIDTHS.get(kind, ()) and M % ROWLIN_ROWS == 0 and M >= ROWLIN_MIN_ROWS and N > 0 and N % 64 == 0


def rowlin(x, w, bias, *, eps=1e-5, geglu=False, ln=True, out=None, gn=None):
    """LayerNorm(x) w^T + bias (geglu: value * gelu(gate) of the interleaved rows) with the rows resident in registers (lr_rowlin_f16):
    the wide short-K projections of a C = 640 block.  w / bias: packing.fold_layernorm (GEGLU: in packing.geglu_perm row order)."""
    lib = _lib.load()
    _chk16(x, "x")
    M, C = x.shape
    N = w.shape[0]
    assert C in (320, 640, 1280) and M % (ROWLIN_ROWS * (2 if C == 320 else 1)) == 0 and N % 64 == 0, (M, C, N)      # (1280: developer builds only; 320: GEGLU only)
    assert w.dtype == x.dtype and w.is_contiguous() and w.shape == (N, C) and bias.dtype == torch.float32 and bias.numel() == N
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, n_out, device=x.device, dtype=x.dtype)
    assert out.shape == (M, n_out) and out.stride(1) == 1 and out.dtype == x.dtype
    a = RowlinArgs()
    a.x, a.w, a.bias, a.out = _p(x), _p(w), _p(bias), _p(out)
    a.M, a.C, a.N, a.ld_out, a.geglu, a.ln_eps, a.ln = M, C, N, out.stride(0), int(bool(geglu)), float(eps), int(bool(ln))
    a.gn_part = a.gn_gamma = a.gn_beta = 0
    a.gn_chunks = a.gn_hw = 0
    a.gn_eps = 0.0
    if gn is not None:      # (gp, chunks, HW, gamma, beta, eps): GroupNorm of the raw rows on the way in, see stin_block
        gp, chunks, hw, gamma, beta, geps = gn
        assert not ln and not geglu and gp.dtype == torch.float32 and gp.is_contiguous() and gp.shape == (M // hw, chunks, 32, 2) and hw % ROWLIN_ROWS == 0
        assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
        a.gn_part, a.gn_gamma, a.gn_beta, a.gn_chunks, a.gn_hw, a.gn_eps = _p(gp), _p(gamma), _p(beta), int(chunks), int(hw), float(geps)
    _lib.check(_fn(lib, "lr_rowlin_f16", x.dtype)(a, _stream()), "rowlin")
    return out


FFN_C, FFN_ROWS, FFN_MAX_H = 320, 128, 2048


def ffn_ok(M, C, H):
    """Shapes lr_ffn_block_f16 takes (everything else keeps the GEGLU GEMM -> Linear GEMM path)."""
    return C == FFN_C and M % FFN_ROWS == 0 and H % 64 == 0 and 0 < H <= FFN_MAX_H


def ffn_block(x, w1, b1, w2, b2, *, eps, want_stats=False, out=None, post=None, want_gn_stats=False, gn_hw=0):
    """x + W2 (u * gelu(g)) + b2 with [u | g] = LayerNorm(x) W1^T + b1, in one launch (lr_ffn_block_f16).
    w1 / b1: LayerNorm-folded GEGLU projection in the interleaved [u16 | g16] row order (packing.pack_geglu / fold_layernorm);
    w2: second Linear as packing.pack_pieces ([H / 64, C, 64]).  post = (Wp pieces, bp, x_in): out = x3 Wp^T + bp + x_in as well
    (SpatialTransformer.proj_out + outer residual).  Returns out [M, C]; (out, stats [M, 2, 2]) with want_stats; (out, (column sums
    [M / 128, C, 2], 128)) with want_gn_stats (needs post; takes precedence over want_stats)."""
    lib = _lib.load()
    _chk16(x, "x")
    M, C = x.shape
    H = w2.shape[0] * 64
    assert ffn_ok(M, C, H), (M, C, H)
    assert w1.dtype == x.dtype and w1.is_contiguous() and w1.shape == (2 * H, C)
    assert w2.dtype == x.dtype and w2.is_contiguous() and w2.shape == (H // 64, C, 64)
    assert b1.dtype == torch.float32 and b1.numel() == 2 * H and b2.dtype == torch.float32 and b2.numel() == C
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous() and out.device == x.device
    stats = torch.empty(M, 2, 2, device=x.device, dtype=torch.float32) if want_stats else None      # one partial per wave of a pair
    a = FfnArgs()
    a.x, a.out, a.w1, a.b1, a.w2, a.b2, a.stats_out = _p(x), _p(out), _p(w1), _p(b1), _p(w2), _p(b2), _p(stats)
    a.M, a.C, a.H, a.ln_eps = M, C, H, float(eps)
    a.post_w = a.post_b = a.post_resid = a.gn_stats_out = 0
    a.gn_group_out, a.gn_hw = 0, 0
    gstats = None
    if post is not None:
        # (Wp pieces [5, C, 64], bp, x_in): out = (x + ff(LN x)) Wp^T + bp + x_in in the same launch (SpatialTransformer.proj_out)
        pw_, pb_, pr_ = post
        _chk16(pr_, "post_resid")
        assert pw_.dtype == x.dtype and pw_.is_contiguous() and pw_.shape == (C // 64, C, 64) and pr_.shape == (M, C)
        assert pb_.dtype == torch.float32 and pb_.numel() == C
        a.post_w, a.post_b, a.post_resid = _p(pw_), _p(pb_), _p(pr_)
        if want_gn_stats:
            gp, chunks = None, 0
            if GN_GROUPS and gn_hw and gn_hw % FFN_ROWS == 0 and M % gn_hw == 0:
                chunks = gn_hw // FFN_ROWS
                gp = torch.empty(M // gn_hw, chunks, 32, 2, device=x.device, dtype=torch.float32)
                a.gn_group_out, a.gn_hw = _p(gp), gn_hw
            gstats = (torch.empty(M // FFN_ROWS, C, 2, device=x.device, dtype=torch.float32), FFN_ROWS, gp, chunks)
            a.gn_stats_out = _p(gstats[0])
    else:
        assert not want_gn_stats
    _lib.check(_fn(lib, "lr_ffn_block_f16", x.dtype)(a, _stream()), "ffn_block")
    if want_gn_stats:
        return out, gstats
    return (out, stats) if want_stats else out


def softmax_rows(s, scale, out=None):
    """p = softmax(scale * s, dim=-1) for materialised fp16 logits s [M, N] (VAE AttnBlock); in place when out is s."""
    lib = _lib.load()
    _chk16(s, "s")
    M, N = s.shape
    if out is None:
        out = torch.empty_like(s)
    _lib.check(lib.lr_softmax_rows_f16(_p(s), _p(out), M, N, float(scale), _stream()), "softmax_rows")
    return out


def mv_gather(x, b, v, s):
    lib = _lib.load()
    _chk16(x, "x")
    C = x.shape[-1]
    seq = torch.empty(b * (v + 1) * s * s, C, device=x.device, dtype=x.dtype)
    _lib.check(_fn(lib, "lr_mv_gather", x.dtype)(_p(x), _p(seq), b, v, s, C, _stream()), "mv_gather")
    return seq


def row_copy(jobs):
    """Up to four row-copy jobs in one launch (lr_row_copy).  job = dict(src, dst, row_bytes, n_rows[, src_off, dst_off, src_idx,
    dst_idx]): tensors with contiguous rows (pitch = stride(0) in bytes); index tables int32 device tensors or None (identity)."""
    lib = _lib.load()
    assert 1 <= len(jobs) <= 4
    arr = (_lib.RowCopyJob * len(jobs))()
    for a, j in zip(arr, jobs):
        src, dst = j["src"], j["dst"]
        assert src.is_cuda and dst.is_cuda and src.stride(-1) == 1 and dst.stride(-1) == 1
        a.src, a.src_pitch, a.src_off = src.data_ptr(), src.stride(0) * src.element_size(), int(j.get("src_off", 0))
        a.dst, a.dst_pitch, a.dst_off = dst.data_ptr(), dst.stride(0) * dst.element_size(), int(j.get("dst_off", 0))
        a.row_bytes, a.n_rows = int(j["row_bytes"]), int(j["n_rows"])
        si, di = j.get("src_idx"), j.get("dst_idx")
        for t_ in (si, di):
            assert t_ is None or (t_.dtype == torch.int32 and t_.is_cuda and t_.is_contiguous() and t_.numel() >= a.n_rows)
        a.src_idx, a.dst_idx = _p(si), _p(di)
    _lib.check(lib.lr_row_copy(arr, len(jobs), _stream()), "row_copy")


def mv_scatter(seq, b, v, s):
    lib = _lib.load()
    _chk16(seq, "seq")
    C = seq.shape[-1]
    x = torch.empty(b * v * s * 2 * s, C, device=seq.device, dtype=seq.dtype)
    _lib.check(_fn(lib, "lr_mv_scatter", seq.dtype)(_p(seq), _p(x), b, v, s, C, _stream()), "mv_scatter")
    return x


def ddim_cfg_step(x, eps, noise, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at):
    """x [B,...] fp32; eps [2B,...] fp16|fp32 (uncond first); returns (x_prev, pred_x0) fp32."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous()
    assert eps.numel() == 2 * x.numel() and eps.dtype in (torch.float16, torch.bfloat16, torch.float32)
    if noise is not None:
        noise = noise.float().contiguous()
    x_prev = torch.empty_like(x)
    pred = torch.empty_like(x)
    _lib.check(_fn(lib, "lr_ddim_cfg_step", torch.bfloat16 if eps.dtype == torch.bfloat16 else torch.float16)(_p(x), _p(eps), int(eps.dtype == torch.float32), _p(noise), _p(x_prev), _p(pred),
                                    x.numel(), float(cfg_scale), float(a_t), float(a_prev), float(sigma_t),
                                    float(sqrt_one_minus_at), _stream()), "ddim_cfg_step")
    return x_prev, pred
