"""ctypes binding of libleftrefill_hip.so (the C ABI declared in include/leftrefill_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module raises, and every
op raises RuntimeError on a non-zero return code.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# LEFTREFILL_LIB_PATH: developer override (same-box A/B of two builds of the library)
LIB_PATH = os.environ.get("LEFTREFILL_LIB_PATH") or os.path.join(HERE, "lib", "libleftrefill_hip.so")
ABI_VERSION = 26

c_void_p, c_int, c_float, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64


class GemmArgs(ctypes.Structure):
    """struct lr_gemm_args (include/leftrefill_hip.h)."""
    _fields_ = [
        ("p1", c_void_p), ("C1", ctypes.c_int32),
        ("p2", c_void_p), ("C2", ctypes.c_int32),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("Hs", ctypes.c_int32), ("Ws", ctypes.c_int32),
        ("taps", ctypes.c_int32), ("stride", ctypes.c_int32), ("up", ctypes.c_int32),
        ("asym", ctypes.c_int32),
        ("wt", c_void_p), ("N", ctypes.c_int32),
        ("bias", c_void_p),
        ("rowvec", c_void_p), ("ld_rowvec", ctypes.c_int32),
        ("resid", c_void_p), ("ld_resid", ctypes.c_int32),
        ("out", c_void_p), ("ld_out", ctypes.c_int32),
        ("geglu", ctypes.c_int32),
        ("tile_n", ctypes.c_int32),
        ("tile_m", ctypes.c_int32),
        ("splits", ctypes.c_int32),
        ("workspace", c_void_p), ("workspace_bytes", ctypes.c_int64),
        ("ln_stats", c_void_p), ("ln_parts", ctypes.c_int32), ("ln_eps", ctypes.c_float),
        ("ln_colsum", c_void_p),
        ("stats_out", c_void_p),
        ("gn_stats_out", c_void_p),
        ("dtype", ctypes.c_int32),
        ("pipe", ctypes.c_int32),
        ("gn_group_out", c_void_p), ("gn_hw", ctypes.c_int32),
        ("wt_bstride", ctypes.c_int32), ("bias_bstride", ctypes.c_int32),
        ("wt_pm", ctypes.c_int32),
        ("skip1", c_void_p), ("skip2", c_void_p), ("Cs1", ctypes.c_int32), ("Cs2", ctypes.c_int32),
        ("splitk_mode", ctypes.c_int32),
    ]


class AttnBwdArgs(ctypes.Structure):
    """struct lr_attn_bwd_args (include/leftrefill_hip.h)."""
    _fields_ = ([(n, c_void_p) for n in ("q", "k", "v", "o", "dout", "qt", "kt", "dot", "lse", "dsum", "dq", "dk", "dv")] +
                [(n, ctypes.c_int32) for n in ("ldq", "ldk", "ldv", "ldo", "lddo", "ld_qt", "ld_kt", "lddq", "lddk", "lddv",
                                               "B", "heads", "Nq", "Nkv")] + [("scale", ctypes.c_float)])


class XattnArgs(ctypes.Structure):
    """struct lr_xattn_args (include/leftrefill_hip.h)."""
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("wq", c_void_p), ("bq", c_void_p), ("k", c_void_p), ("ldk", ctypes.c_int32),
                ("vt", c_void_p), ("wo", c_void_p), ("bo", c_void_p), ("stats_out", c_void_p),
                ("M", ctypes.c_int32), ("HW", ctypes.c_int32), ("C", ctypes.c_int32), ("heads", ctypes.c_int32),
                ("Lc", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("scale", ctypes.c_float),
                ("pre_a", c_void_p), ("pre_w", c_void_p), ("pre_b", c_void_p)]


class RowCopyJob(ctypes.Structure):
    """struct lr_row_copy_job (include/leftrefill_hip.h)."""
    _fields_ = [("src", c_void_p), ("src_pitch", ctypes.c_int64), ("src_off", ctypes.c_int64),
                ("dst", c_void_p), ("dst_pitch", ctypes.c_int64), ("dst_off", ctypes.c_int64),
                ("row_bytes", ctypes.c_int32), ("n_rows", ctypes.c_int32), ("src_idx", c_void_p), ("dst_idx", c_void_p)]


class StinArgs(ctypes.Structure):
    """struct lr_stin_args (include/leftrefill_hip.h)."""
    _fields_ = [("x", c_void_p), ("wp", c_void_p), ("bp", c_void_p), ("wqkv", c_void_p), ("bqkv", c_void_p), ("x1", c_void_p),
                ("qkv", c_void_p), ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("NQ", ctypes.c_int32), ("ld_qkv", ctypes.c_int32),
                ("ln_eps", ctypes.c_float), ("gn_part", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p),
                ("gn_chunks", ctypes.c_int32), ("gn_hw", ctypes.c_int32), ("gn_eps", ctypes.c_float)]


class RowlinArgs(ctypes.Structure):
    """struct lr_rowlin_args (include/leftrefill_hip.h)."""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p), ("M", ctypes.c_int32), ("C", ctypes.c_int32),
                ("N", ctypes.c_int32), ("ld_out", ctypes.c_int32), ("geglu", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("ln", ctypes.c_int32),
                ("gn_part", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_chunks", ctypes.c_int32), ("gn_hw", ctypes.c_int32),
                ("gn_eps", ctypes.c_float)]


class FfnArgs(ctypes.Structure):
    """struct lr_ffn_args (include/leftrefill_hip.h)."""
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
                ("stats_out", c_void_p), ("M", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("ln_eps", ctypes.c_float),
                ("post_w", c_void_p), ("post_b", c_void_p), ("post_resid", c_void_p), ("gn_stats_out", c_void_p),
                ("gn_group_out", c_void_p), ("gn_hw", ctypes.c_int32)]


# symbol -> argtypes; every function returns int
SIGNATURES = {
    "lr_abi_version": [],
    "lr_nchw_f32_to_nhwc_f16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_nhwc_f16_to_nchw": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_groupnorm_stats": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "lr_groupnorm_apply": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                           c_int, c_void_p, c_void_p],
    "lr_layernorm_bwd": [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    "lr_groupnorm_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                         c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "lr_layernorm_bwd_res": [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    "lr_groupnorm_bwd_res": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                             c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "lr_mv_gather_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_mv_scatter_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_geglu_fwd": [c_void_p, c_void_p, c_int, c_int, c_void_p],
    "lr_geglu_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "lr_sumpool2x2": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_softmax_rows_f16": [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "lr_layernorm": [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    "lr_timestep_embedding": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "lr_linear_small_m": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_void_p],
    "lr_gemm_workspace_bytes": [ctypes.POINTER(GemmArgs)],
    "lr_gemm_stats_parts": [ctypes.POINTER(GemmArgs)],
    "lr_gemm_gn_rows": [ctypes.POINTER(GemmArgs)],
    "lr_gemm_gn_group_chunks": [ctypes.POINTER(GemmArgs)],
    "lr_gn_conv_out_f16": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p,
                           c_int, c_void_p, c_void_p],
    "lr_groupnorm_finalize": [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "lr_groupnorm_apply_n": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float,
                             c_int, c_void_p, c_void_p],
    "lr_gemm_plan": [ctypes.POINTER(GemmArgs), ctypes.POINTER(ctypes.c_int32)],
    "lr_gemm_conv_f16": [ctypes.POINTER(GemmArgs), c_void_p],
    "lr_attention_f16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                         c_int, c_float, c_void_p],
    "lr_attention_vt_f16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                         c_int, c_float, c_void_p],
    "lr_attention_causal_f16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                c_void_p],
    "lr_attention_lse_f16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                             c_int, c_float, c_void_p],
    "lr_attention_bwd_f16": [ctypes.POINTER(AttnBwdArgs), c_void_p],
    "lr_transpose_v_f16": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_xattn_block_f16": [ctypes.POINTER(XattnArgs), c_void_p],
    "lr_xattn_pack_vt_f16": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "lr_ffn_block_f16": [ctypes.POINTER(FfnArgs), c_void_p],
    "lr_stin_block_f16": [ctypes.POINTER(StinArgs), c_void_p],
    "lr_rowlin_f16": [ctypes.POINTER(RowlinArgs), c_void_p],
    "lr_mv_gather": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_row_copy": [c_void_p, c_int, c_void_p],
    "lr_gemm_splitk_timeouts": [],
    "lr_mv_scatter": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "lr_ddim_cfg_step": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                         c_float, c_float, c_void_p],
}

# bfloat16 twins (include/leftrefill_hip.h, last section): same argument lists as the fp16 entry points
BF16_TWINS = ["lr_groupnorm_stats", "lr_groupnorm_apply", "lr_groupnorm_apply_n", "lr_layernorm", "lr_layernorm_bwd",
              "lr_groupnorm_bwd", "lr_layernorm_bwd_res", "lr_groupnorm_bwd_res", "lr_nchw_f32_to_nhwc_f16", "lr_nhwc_f16_to_nchw", "lr_timestep_embedding", "lr_linear_small_m",
              "lr_mv_gather", "lr_mv_scatter", "lr_ddim_cfg_step", "lr_geglu_fwd", "lr_geglu_bwd", "lr_sumpool2x2",
              "lr_mv_gather_bwd", "lr_mv_scatter_bwd", "lr_attention_f16", "lr_attention_causal_f16", "lr_attention_lse_f16",
              "lr_attention_vt_f16", "lr_transpose_v_f16", "lr_attention_bwd_f16", "lr_xattn_block_f16",
              "lr_xattn_pack_vt_f16", "lr_ffn_block_f16", "lr_stin_block_f16", "lr_rowlin_f16", "lr_gn_conv_out_f16"]


def twin(name):
    """fp16 entry point -> its bfloat16 twin."""
    return (name[:-4] if name.endswith("_f16") else name) + "_bf16"


for _n in BF16_TWINS:
    SIGNATURES[twin(_n)] = SIGNATURES[_n]

# entry points of developer builds only (-DLR_DEV_VARIANTS, include/leftrefill_hip.h last section): bound when present
_FOLD_SIG = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
DEV_SIGNATURES = {"lr_dev_set": [ctypes.c_char_p, c_int], "lr_dev_unset": [ctypes.c_char_p],
                  "lr_gn_fold_weights_f16": _FOLD_SIG, "lr_gn_fold_weights_bf16": _FOLD_SIG}

_lib = None


def fn(lib, name, dtype):
    """The entry point `name` for 16-bit element type `dtype` (torch.float16 | torch.bfloat16)."""
    import torch
    if dtype == torch.bfloat16:
        return getattr(lib, twin(name))
    assert dtype == torch.float16, dtype
    return getattr(lib, name)


def load():
    """Load the shared library once; raises if it is missing (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so) and owns the
    # streams we launch on; loading our library afterwards binds its libamdhip64 dependency to that same runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m leftrefill_amd.build` (hipcc --offload-arch=gfx950). "
            "leftrefill_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.argtypes = argtypes
        fn.restype = c_int64 if name == "lr_gemm_workspace_bytes" else c_int
    v = lib.lr_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libleftrefill_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    for name, argtypes in DEV_SIGNATURES.items():
        if hasattr(lib, name):
            f_ = getattr(lib, name)
            f_.argtypes, f_.restype = argtypes, c_int
    _lib = lib
    if dev_variants():      # developer library: its knobs come from the LR_* environment variables of the tools/ A/B scripts
        for k, val in os.environ.items():
            if k.startswith("LR_") and val.lstrip("-").isdigit():
                lib.lr_dev_set(k.encode(), int(val))
    return lib


def dev_variants():
    """True when the loaded library is a developer build (tools/build_variant.sh: -DLR_DEV_VARIANTS) -- the only kind that has knobs."""
    return hasattr(load(), "lr_dev_set")


def dev_set(name, value):
    """Set (int) or clear (None) a developer knob of a developer build; the product library has none (RuntimeError)."""
    lib = load()
    if not hasattr(lib, "lr_dev_set"):
        raise RuntimeError(f"{name}: developer knobs exist only in a -DLR_DEV_VARIANTS build (tools/build_variant.sh)")
    if value is None:
        lib.lr_dev_unset(name.encode())
    else:
        lib.lr_dev_set(name.encode(), int(value))


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "alignment", -3: "unsupported"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {kind} (rc={rc})")
