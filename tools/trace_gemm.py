"""Developer tool: per-phase timeline of gemm_conv_pipe_kernel from in-kernel shader-clock stamps (needs a trace build):

    hipcc ... -DLR_GEMM_TRACE (python tools/trace_gemm.py builds its own copy of the library under /tmp)

Stamps (thread 0 of each block): 0 start of staging, 1 prologue stages issued, 2 first stage landed + barrier,
3 main loop done, 4 LayerNorm rows ready, 5 epilogue issued, 6 epilogue's memory ops retired.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_trace_lib():
    from leftrefill_amd import build
    out = "/tmp/lr_trace"
    os.makedirs(out, exist_ok=True)
    objs = []
    for s in build.SOURCES:
        o = os.path.join(out, s.replace(".hip", ".o"))
        subprocess.run([build.hipcc()] + build.FLAGS + build.EXTRA.get(s, []) + ["-DLR_GEMM_TRACE", "-c",
                       os.path.join(build.CSRC, s), "-o", o], check=True)
        objs.append(o)
    lib = os.path.join(out, "libleftrefill_hip.so")
    subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def main():
    lib_path = build_trace_lib()
    from leftrefill_amd import _lib
    _lib.LIB_PATH = lib_path
    import torch
    from leftrefill_amd import ops
    import tools.bench_shapes as bs
    lib = _lib.load()
    lib.lr_gemm_set_trace.argtypes = [ctypes.c_void_p]
    lib.lr_gemm_set_trace.restype = None
    dev = torch.device("cuda:0")
    tiles = [(256, 320, 0), (256, 160, 0), (256, 256, 0)]
    argv = sys.argv[1:]
    splits = 1
    if argv and argv[0].startswith("--splits="):      # split-K factor of every launch (blockIdx.y slices; stamps per (slice, tile))
        splits = int(argv.pop(0)[9:])
    if argv and argv[0].startswith("--tiles="):      # e.g. --tiles=128x160x4,256x128
        tiles = [tuple((list(map(int, t_.split("x"))) + [0])[:3]) for t_ in argv.pop(0)[8:].split(",")]
    only = argv or ["l0 KC", "l0 qkv", "l0 geglu", "l0 conv3x3 resid"]
    for name, M, N, K, taps, fl in bs.SHAPES:
        if not any(o in name for o in only):
            continue
        sets, launch = bs.make_case(M, N, K, taps, fl, dev, 6)
        for tm, tn, stg in tiles:
            if fl.get("geglu") and tn == 160:
                continue
            trace = torch.zeros(8192 * 8, device=dev, dtype=torch.int64)
            for i in range(4):
                launch(sets[i], tm, tn, splits, stg)       # warm
            torch.cuda.synchronize()
            lib.lr_gemm_set_trace(trace.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(sets[4], tm, tn, splits, stg)
            e1.record()
            torch.cuda.synchronize()
            lib.lr_gemm_set_trace(None)
            tr = trace.reshape(-1, 8).cpu()
            nb = int((tr[:, 0] != 0).sum())
            tr = tr[:nb].double()
            t0 = tr[:, 0].min()
            ph = [(tr[:, k + 1] - tr[:, k]).mean().item() for k in range(6)]
            span = (tr[:, 6].max() - t0).item()
            start_spread = (tr[:, 0].max() - t0).item()
            end = (tr[:, 6] - t0)
            print(f"{name:24s} tile {tm}x{tn}/{stg} blocks {nb:5d} kernel {1e3 * e0.elapsed_time(e1):7.1f} us | cycles: span {span:9.0f} "
                  f"start-spread {start_spread:8.0f} | issue {ph[0]:7.0f} first-land {ph[1]:7.0f} loop {ph[2]:7.0f} ln {ph[3]:6.0f} "
                  f"epi-issue {ph[4]:7.0f} drain {ph[5]:7.0f} | block end min/mean/max {end.min().item():8.0f} {end.mean().item():8.0f} {end.max().item():8.0f}",
                  flush=True)
        del sets, launch
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
