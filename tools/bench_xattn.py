"""Micro-benchmark: fused cross-attention block (lr_xattn_block_f16) vs the to_q -> attention -> to_out launches it replaces,
at the level-0 shape of the configs[1] UNet step (M = 8 x 8192 rows, C = 320, 77 context tokens).  MI355X.

    python tools/bench_xattn.py [--B 8] [--L 8192]

"cold": every launch of a timed sequence uses another buffer set (rotating over > 256 MB, the Infinity Cache size) -- what
the kernels see inside the UNet step; "hot": one buffer set re-launched.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops, packing  # noqa: E402


def time_seq(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--L", type=int, default=8192)
    ap.add_argument("--Lc", type=int, default=77)
    ap.add_argument("--C", type=int, default=320, help="320 (level 0, default L 8192) | 640 (level 1: pass --L 2048)")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    C, heads, B, L, Lc = a.C, a.C // 64, a.B, a.L, a.Lc
    M = B * L
    g = torch.Generator(device="cpu").manual_seed(0)
    wq = torch.randn(C, C, generator=g) / C ** 0.5
    wk = torch.randn(C, 1024, generator=g) / 32
    wv = torch.randn(C, 1024, generator=g) / 32
    wo = torch.randn(C, C, generator=g) / C ** 0.5
    bo = torch.randn(C, generator=g).to(d)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wqf, bqf, cs = [t.to(d) for t in packing.fold_layernorm(wq, None, gamma, beta)]
    xk_w, xwo = [t.to(d) for t in packing.pack_xattn(wk, wo)]
    kvw = torch.cat([wk, wv], 0).half().to(d)
    wo_d = wo.half().to(d)
    ctx = torch.randn(B * Lc, 1024, generator=g).half().to(d)
    kv = ops.gemm_conv(ctx, kvw, B=1, H=1, W=B * Lc, taps=1)
    kx = ops.gemm_conv(ctx, xk_w, B=1, H=1, W=B * Lc, taps=1)
    vt = ops.xattn_pack_vt(kv[:, C:], B, heads, Lc)
    nsets = max(2, int(600e6 / (M * C * 2 * 2)) + 1)
    xs = [torch.randn(M, C, device=d).half() for _ in range(nsets)]
    sts = []
    for x in xs:
        xf = x.float()
        sts.append(torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(M, 1, 2).contiguous())
    outs = [torch.empty(M, C, device=d, dtype=torch.float16) for _ in range(nsets)]

    def fused(i):
        return ops.xattn_block(xs[i], wqf, bqf, kx, vt, xwo, bo, HW=L, heads=heads, Lc=Lc, eps=1e-5, scale=0.125, want_stats=True,
                               out=outs[i])

    def plain(i):
        q = ops.gemm_conv(xs[i], wqf, B=1, H=1, W=M, taps=1, bias=bqf, ln=(sts[i], 1e-5, cs))
        o = ops.attention_q_kv(q, kv, B, heads, L, Lc, 0.125)
        return ops.gemm_conv(o, wo_d, B=1, H=1, W=M, taps=1, bias=bo, resid=xs[i], want_stats=True, out=outs[i])

    wo1 = (torch.randn(C, C, generator=g) / C ** 0.5).half().to(d)
    bo1 = torch.randn(C, generator=g).to(d)
    wq_pi = wqf[:, packing.xattn_perm(C).to(d)].contiguous()
    a_s = [torch.randn(M, C, device=d).half() for _ in range(nsets)]

    def fused_pre(i):      # attn1's out-projection + residual in front, same launch
        return ops.xattn_block(xs[i], wq_pi, bqf, kx, vt, xwo, bo, HW=L, heads=heads, Lc=Lc, eps=1e-5, scale=0.125, want_stats=True,
                               out=outs[i], pre=(a_s[i], wo1, bo1))

    def plain_pre(i):
        x1, st1 = ops.gemm_conv(a_s[i], wo1, B=1, H=1, W=M, taps=1, bias=bo1, resid=xs[i], want_stats=True)
        q = ops.gemm_conv(x1, wqf, B=1, H=1, W=M, taps=1, bias=bqf, ln=(st1, 1e-5, cs))
        o = ops.attention_q_kv(q, kv, B, heads, L, Lc, 0.125)
        return ops.gemm_conv(o, wo_d, B=1, H=1, W=M, taps=1, bias=bo, resid=x1, want_stats=True, out=outs[i])

    res = {}
    for name, fn in (("fused", fused), ("three launches", plain), ("fused + pre", fused_pre), ("four launches", plain_pre)):
        fn(0)
        torch.cuda.synchronize()
        cold = min(time_seq(lambda i: fn(i % nsets), nsets * 2) for _ in range(3))
        hot = min(time_seq(lambda i: fn(0), 8) for _ in range(3))
        res[name] = (cold, hot)
    o1 = fused(0)[0].float()
    o2 = plain(0)[0].float()
    flops = 2.0 * M * C * C * 2 + 4.0 * M * Lc * C
    byt = 3.0 * M * C * 2
    print(f"xattn block M={M} C={C} Lc={Lc}: max |fused - plain| = {(o1 - o2).abs().max().item():.3e}")
    for name, (cold, hot) in res.items():
        print(f"  {name:15s} cold {cold:7.1f} us  hot {hot:7.1f} us   ({flops / cold / 1e6:6.0f} TFLOP/s, {byt / cold / 1e6:5.2f} TB/s of x + resid... + out)")


if __name__ == "__main__":
    main()
