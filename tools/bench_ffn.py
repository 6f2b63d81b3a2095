"""Micro-benchmark: fused feed-forward block (lr_ffn_block_f16) vs the LayerNorm-folded GEGLU GEMM + second Linear GEMM it replaces,
level-0 shape of the configs[1] UNet step (M = 8 x 8192 rows, C = 320, H = 1280).  MI355X.  cold = rotating buffer sets (> 256 MB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops, packing  # noqa: E402
from tools.bench_xattn import time_seq  # noqa: E402


def main():
    d = torch.device("cuda:0")
    C, H, M = 320, 1280, 65536
    g = torch.Generator(device="cpu").manual_seed(0)
    w1 = torch.randn(2 * H, C, generator=g) / C ** 0.5
    b1 = 0.1 * torch.randn(2 * H, generator=g)
    w2 = torch.randn(C, H, generator=g) / H ** 0.5
    b2 = (0.1 * torch.randn(C, generator=g)).to(d)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wf, bf, cs = packing.fold_layernorm(w1, b1, gamma, beta)
    perm = packing.geglu_perm(H)
    w1p, b1p, csp = wf[perm].contiguous().to(d), bf[perm].contiguous().to(d), cs[perm].contiguous().to(d)
    w2h = w2.half().to(d)
    w2x = packing.pack_pieces(w2).to(d)
    nsets = max(2, int(600e6 / (M * C * 2 * 2)) + 1)
    xs = [torch.randn(M, C, device=d).half() for _ in range(nsets)]
    sts = []
    for x in xs:
        xf = x.float()
        sts.append(torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(M, 1, 2).contiguous())
    outs = [torch.empty(M, C, device=d, dtype=torch.float16) for _ in range(nsets)]

    def fused(i):
        return ops.ffn_block(xs[i], w1p, b1p, w2x, b2, eps=1e-5, out=outs[i])

    def plain(i):
        h = ops.gemm_conv(xs[i], w1p, B=1, H=1, W=M, taps=1, bias=b1p, geglu=True, ln=(sts[i], 1e-5, csp))
        return ops.gemm_conv(h, w2h, B=1, H=1, W=M, taps=1, bias=b2, resid=xs[i], out=outs[i])

    wp = (torch.randn(C, C, generator=g) / C ** 0.5)
    wp_h, wp_x = wp.half().to(d), packing.pack_pieces(wp).to(d)
    bp = (0.1 * torch.randn(C, generator=g)).to(d)
    xins = [torch.randn(M, C, device=d).half() for _ in range(nsets)]

    def fused_post(i):      # proj_out + x_in behind the block, same launch, with the GroupNorm statistics of the output
        return ops.ffn_block(xs[i], w1p, b1p, w2x, b2, eps=1e-5, out=outs[i], post=(wp_x, bp, xins[i]), want_gn_stats=True)

    def fused_then_gemm(i):
        y = ops.ffn_block(xs[i], w1p, b1p, w2x, b2, eps=1e-5)
        return ops.gemm_conv(y, wp_h, B=1, H=1, W=M, taps=1, bias=bp, resid=xins[i], out=outs[i], want_gn_stats=True)

    res = {}
    for name, fn in (("fused", fused), ("two GEMMs", plain), ("fused + post", fused_post), ("fused, proj_out", fused_then_gemm)):
        fn(0)
        torch.cuda.synchronize()
        cold = min(time_seq(lambda i: fn(i % nsets), nsets * 2) for _ in range(3))
        hot = min(time_seq(lambda i: fn(0), 8) for _ in range(3))
        res[name] = (cold, hot)
    o1, o2 = fused(0).float(), plain(0).float()
    flops = 2.0 * M * C * 2 * H + 2.0 * M * H * C
    print(f"ffn block M={M} C={C} H={H}: max |fused - plain| = {(o1 - o2).abs().max().item():.3e} at |out| {o2.abs().max().item():.2f}")
    for name, (cold, hot) in res.items():
        print(f"  {name:12s} cold {cold:7.1f} us  hot {hot:7.1f} us   ({flops / cold / 1e6:6.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
