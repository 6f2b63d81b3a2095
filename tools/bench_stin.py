"""Micro-benchmark: fused SpatialTransformer entry (lr_stin_block_f16: proj_in + LayerNorm + q|k|v projection) vs the two GEMMs it replaces,
at the level-0 shape of the configs[1] UNet step (M = 8 x 8192 rows, C = 320).  MI355X.   python tools/bench_stin.py [--M 65536]

"cold": every launch of a timed sequence uses another buffer set (rotating over > 256 MB); "hot": one buffer set re-launched."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops, packing  # noqa: E402
from tools.bench_xattn import time_seq  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=65536)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    C, NQ, M = 320, 960, a.M
    g = torch.Generator(device="cpu").manual_seed(0)
    wp = (torch.randn(C, C, generator=g) / C ** 0.5).half().to(d)
    bp = torch.randn(C, generator=g).to(d)
    wq = torch.randn(NQ, C, generator=g) / C ** 0.5
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wf, bf, cs = [t.to(d) for t in packing.fold_layernorm(wq, None, gamma, beta)]
    nsets = max(2, int(600e6 / (M * (2 * C + NQ) * 2)) + 1)
    xs = [torch.randn(M, C, device=d).half() for _ in range(nsets)]
    x1s = [torch.empty(M, C, device=d, dtype=torch.float16) for _ in range(nsets)]
    qs = [torch.empty(M, NQ, device=d, dtype=torch.float16) for _ in range(nsets)]

    def fused(i):
        return ops.stin_block(xs[i], wp, bp, wf, bf, eps=1e-5, out=x1s[i], qkv_out=qs[i])

    def plain(i):
        y1, st = ops.gemm_conv(xs[i], wp, B=1, H=1, W=M, taps=1, bias=bp, want_stats=True, out=x1s[i])
        return ops.gemm_conv(y1, wf, B=1, H=1, W=M, taps=1, bias=bf, ln=(st, 1e-5, cs), out=qs[i])

    res = {}
    for name, fn in (("fused", fused), ("two GEMMs", plain)):
        fn(0)
        torch.cuda.synchronize()
        cold = min(time_seq(lambda i: fn(i % nsets), nsets * 2) for _ in range(3))
        hot = min(time_seq(lambda i: fn(0), 8) for _ in range(3))
        res[name] = (cold, hot)
    flops = 2.0 * M * C * (C + NQ)
    byt = 2.0 * M * (2 * C + NQ)
    for name, (cold, hot) in res.items():
        print(f"  {name:10s} cold {cold:7.1f} us  hot {hot:7.1f} us   ({flops / cold / 1e6:6.0f} TFLOP/s, {byt / cold / 1e6:5.2f} TB/s of x + x1 + qkv)")


if __name__ == "__main__":
    main()
