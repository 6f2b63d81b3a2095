"""Micro-benchmark: row-resident LayerNorm + Linear at C = 640 (lr_rowlin_f16) vs the LayerNorm-folded tiled GEMMs it replaces, at the
level-1 shapes of the configs[1] UNet step (M = 8 x 2048 rows): q|k|v (N = 1920) and GEGLU (N = 5120).  MI355X.   python tools/bench_rowlin.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops, packing  # noqa: E402
from tools.bench_xattn import time_seq  # noqa: E402


def main():
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    from leftrefill_amd import _lib
    shapes = [(640, 16384, False, 1920), (640, 16384, True, 5120), (640, 16384, False, 640)]
    if _lib.dev_variants():      # the level-0 / level-2 instances exist in developer builds only (measured, lost)
        shapes += [(320, 65536, True, 2560), (1280, 4096, False, 1280), (1280, 4096, False, 3840), (1280, 4096, True, 10240)]
    for C, M, geglu, N in shapes:
        gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        w = torch.randn(N, C, generator=g) / C ** 0.5
        b = torch.randn(N, generator=g)
        wf, bf, cs = packing.fold_layernorm(w, b, gamma, beta)
        if geglu:
            perm = packing.geglu_perm(N // 2)
            wf, bf, cs = wf[perm].contiguous(), bf[perm].contiguous(), cs[perm].contiguous()
        wf, bf, cs = wf.to(d), bf.to(d), cs.to(d)
        n_out = N // 2 if geglu else N
        nsets = max(2, int(600e6 / (M * (C + n_out) * 2)) + 1)
        xs = [torch.randn(M, C, device=d).half() for _ in range(nsets)]
        sts = []
        for x in xs:
            xf = x.float()
            sts.append(torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(M, 1, 2).contiguous())
        outs = [torch.empty(M, n_out, device=d, dtype=torch.float16) for _ in range(nsets)]
        fused = lambda i: ops.rowlin(xs[i], wf, bf, eps=1e-5, geglu=geglu, out=outs[i])
        plain = lambda i: ops.gemm_conv(xs[i], wf, B=1, H=1, W=M, taps=1, bias=bf, geglu=geglu, ln=(sts[i], 1e-5, cs), out=outs[i])
        flops = 2.0 * M * C * N
        print(f"M={M} N={N} K={C} geglu={geglu}")
        for name, fn in (("row-resident", fused), ("tiled GEMM", plain)):
            fn(0)
            torch.cuda.synchronize()
            cold = min(time_seq(lambda i: fn(i % nsets), nsets * 2) for _ in range(3))
            hot = min(time_seq(lambda i: fn(0), 8) for _ in range(3))
            print(f"  {name:13s} cold {cold:7.1f} us  hot {hot:7.1f} us   ({flops / cold / 1e6:6.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
