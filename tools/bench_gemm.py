"""Micro-benchmark of lr_gemm_conv_f16 for one shape across tile configurations (MI355X only).

    python tools/bench_gemm.py M N K taps [tile_m tile_n splits [pipe]] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402


GRAPH = False
COLD = 0
COLD_X = False


def run(M, N, K, taps, tm, tn, splits, reps, geglu=False, pipe=0):
    dev = torch.device("cuda:0")
    C = K // taps
    if taps == 9:
        B = 8
        HW = M // B
        W = int((HW * 2) ** 0.5)
        Hh = HW // W
    else:
        B, Hh, W = 1, 1, M
    x = torch.randn(M, C, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
    if COLD > 1:      # COLD distinct weight (and input) copies, visited in turn: every launch streams its weights from HBM like a layer of the step
        ws = [w.clone() for _ in range(COLD)]
        xs = [x.clone() for _ in range(COLD if COLD_X else 1)]
        cnt = [0]

        def f():
            i = cnt[0] % COLD
            cnt[0] += 1
            return ops.gemm_conv(xs[i % len(xs)], ws[i], B=B, H=Hh, W=W, taps=taps, bias=b, out=out, tile_m=tm, tile_n=tn, splits=splits, geglu=geglu, pipe=pipe)
    else:
        f = lambda: ops.gemm_conv(x, w, B=B, H=Hh, W=W, taps=taps, bias=b, out=out, tile_m=tm, tile_n=tn, splits=splits,
                                  geglu=geglu, pipe=pipe)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if GRAPH:      # the launches replayed from a hipGraph: GPU time without the host's per-call cost (~15 us through ctypes)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                f()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        return us, 2.0 * M * N * K / us / 1e6
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dims", type=int, nargs="+")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--geglu", action="store_true")
    ap.add_argument("--graph", action="store_true", help="time a hipGraph of the launches (no host cost per call)")
    ap.add_argument("--cold", type=int, default=0, help="cycle through this many weight copies (> cache capacity: every launch reads HBM)")
    ap.add_argument("--cold-x", action="store_true", help="... and as many input copies")
    a = ap.parse_args()
    GRAPH = a.graph
    COLD, COLD_X = a.cold, a.cold_x
    M, N, K, taps = a.dims[:4]
    ops.AUTOTUNE = False
    if len(a.dims) >= 6:
        cfgs = [(a.dims[4], a.dims[5], a.dims[6] if len(a.dims) > 6 else 0)]
    else:
        cfgs = [(128, 64, 0), (128, 128, 0), (128, 160, 0), (256, 128, 0), (256, 160, 0), (256, 320, 0)]
    pipe = a.dims[7] if len(a.dims) > 7 else 0
    for tm, tn, sp in cfgs:
        try:
            if a.geglu and tn == 160:
                continue
            us, tf = run(M, N, K, taps, tm, tn, sp, a.reps, a.geglu, pipe)
            print(f"M={M} N={N} K={K} taps={taps} tile {tm}x{tn} splits={sp} pipe={pipe}: {us:8.1f} us  {tf:7.1f} TFLOP/s")
        except Exception as e:  # noqa: BLE001
            print(f"tile {tm}x{tn}: {e}")
