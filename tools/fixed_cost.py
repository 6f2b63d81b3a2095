"""Fixed cost of a launch per kernel_table row: the same (M, N, tile plan) with ONE K-step (K = 64, pointwise), i.e. launch + pipeline
fill + the epilogue's residual read / store -- the third term of tools/floor_table.py.  MI355X.  Timed as hipGraph replays (GPU side only).

    python tools/fixed_cost.py BENCH.json > profiles/r06_fixed_cost.json

Attention rows: one 64-key tile; fused-block rows: no 1-step form exists (0)."""
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402
from tools.floor_table import load_line  # noqa: E402


def timed(fn, n=20):
    """us per launch of n launches replayed from ONE hipGraph: the GPU-side cost.  (Round 6's first table timed eager launches: ~10 us per
    row, most of it the host's ctypes call -- the same launches replay in 5.6-7.5 us.)"""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / n)
    return best


def main():
    r = load_line(sys.argv[1])
    d = torch.device("cuda:0")
    out = {}
    for t in r["kernel_table"]:
        s = t["shape"]
        m = re.match(r"gemm (\d+)x(\d+)x(\d+) taps(\d) s(\d) up(\d)( geglu)?", s)
        if m:
            M, N, K, taps = int(m[1]), int(m[2]), int(m[3]), int(m[4])
            geglu = bool(m[7])
            x = torch.randn(M, 64, device=d).half()
            w = torch.randn(N, 64, device=d).half()
            # the plan of the real shape (tile, no split), one K-step
            plan = ops.gemm_plan(M, N, K, taps=taps, stride=int(m[5]), up=int(m[6]), geglu=geglu)
            tm, tn, _sp, pipe = plan
            if pipe == 8:
                pipe = 0      # the halo instance is a 3x3 kernel: its tile's gather instance has the same fill / epilogue shape
            o = torch.empty(M, N // 2 if geglu else N, device=d, dtype=torch.float16)
            try:
                out[s] = round(timed(lambda: ops.gemm_conv(x, w, B=1, H=1, W=M, taps=1, geglu=geglu, out=o, tile_m=tm, tile_n=tn, splits=1, pipe=pipe)), 2)
            except RuntimeError:
                out[s] = 0.0
            continue
        m = re.match(r"attn B(\d+) h(\d+) (\d+)x(\d+)", s)
        if m:
            B, h, Nq = int(m[1]), int(m[2]), int(m[3])
            q = torch.randn(B * Nq, h * 64, device=d).half()
            k = torch.randn(B * 64, h * 64, device=d).half()
            out[s] = round(timed(lambda: ops.attention(q, k, k, B, h, Nq, 64, 0.125)), 2)
            continue
        out[s] = 0.0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
