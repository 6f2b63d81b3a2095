"""Developer tool: per-phase timeline of xattn_block_kernel from in-kernel shader-clock stamps (trace build of the library:
tools/build_variant.sh xatrace -DLR_XATTN_TRACE, run with LEFTREFILL_LIB_PATH=leftrefill_amd/lib/variants/libleftrefill_hip_xatrace.so).

Stamps per (block, wave): 0 start, 1 rows loaded + LayerNorm, then per head h: 2+4h step A starts (after its barrier), 3+4h step A's
MFMAs done, 4+4h step B starts, 5+4h step C starts; 22 main loop done, 23 epilogue's stores retired.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from leftrefill_amd import _lib, ops, packing  # noqa: E402


def main():
    lib = _lib.load()
    lib.lr_xattn_set_trace.argtypes = [ctypes.c_void_p]
    lib.lr_xattn_set_trace.restype = None
    d = torch.device("cuda:0")
    C, heads, B, L, Lc = 320, 5, 8, 8192, 77
    M = B * L
    wq = (torch.randn(C, C) / C ** 0.5)
    wk, wv, wo = torch.randn(C, 1024) / 32, torch.randn(C, 1024) / 32, torch.randn(C, C) / C ** 0.5
    wqf, bqf, _ = [t.to(d) for t in packing.fold_layernorm(wq, None, torch.ones(C), torch.zeros(C))]
    xk_w, xwo = [t.to(d) for t in packing.pack_xattn(wk, wo)]
    ctx = torch.randn(B * Lc, 1024).half().to(d)
    kx = ops.gemm_conv(ctx, xk_w, B=1, H=1, W=B * Lc, taps=1)
    v = ops.gemm_conv(ctx, wv.half().to(d), B=1, H=1, W=B * Lc, taps=1)
    vt = ops.xattn_pack_vt(v, B, heads, Lc)
    bo = torch.zeros(C, device=d)
    xs = [torch.randn(M, C, device=d).half() for _ in range(8)]
    run = lambda x: ops.xattn_block(x, wqf, bqf, kx, vt, xwo, bo, HW=L, heads=heads, Lc=Lc, eps=1e-5, scale=0.125, want_stats=True)
    for x in xs[:4]:
        run(x)
    torch.cuda.synchronize()
    nb = M // 128
    trace = torch.zeros(nb * 8 * 24, device=d, dtype=torch.int64)
    lib.lr_xattn_set_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(xs[5])
    e1.record()
    torch.cuda.synchronize()
    lib.lr_xattn_set_trace(None)
    tr = trace.reshape(nb, 8, 24).cpu().double()
    t0 = tr[:, :, 0].min()
    start = tr[:, 0, 0] - t0
    first = start < start.median()           # first-round blocks (start with the launch) vs second round
    print(f"kernel {1e3 * e0.elapsed_time(e1):.1f} us; span {(tr[:, :, 23].max() - t0).item():.0f} cycles "
          f"(100 MHz-independent shader clock); blocks {nb}; first-round {int(first.sum())}")
    for name, sel in (("first round", first), ("second round", ~first)):
        t = tr[sel]
        seg = lambda a, b_: (t[:, :, b_] - t[:, :, a]).mean().item()
        print(f"[{name}] start {(t[:, :, 0] - t0).mean().item():8.0f}  rows+LN {seg(0, 1):7.0f}  wait first piece {seg(1, 2):7.0f}")
        for h in range(5):
            a, am, b_, c_ = 2 + 4 * h, 3 + 4 * h, 4 + 4 * h, 5 + 4 * h
            nxt = 2 + 4 * (h + 1) if h < 4 else 22
            print(f"   head {h}: A mfma {seg(a, am):6.0f}  A->B wait {seg(am, b_):6.0f}  B (attn) + wait {seg(b_, c_):6.0f}  C + wait {seg(c_, nxt):6.0f}")
        print(f"   epilogue {seg(22, 23):7.0f}   block total {seg(0, 23):8.0f}")


if __name__ == "__main__":
    main()
