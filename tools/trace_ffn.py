"""Developer tool: per-phase timeline of ffn_block_kernel from shader-clock stamps (tools/build_variant.sh fftrace -DLR_FFN_TRACE, run with
LEFTREFILL_LIB_PATH=leftrefill_amd/lib/variants/libleftrefill_hip_fftrace.so).  Stamps per (block, wave): 0 start, 1 rows + LayerNorm,
2+4j / 3+4j / 4+4j / 5+4j = step A1 starts / A1 done / A2 starts / C starts of super-chunk j < 3, 14 main loop done, 15 stores retired."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from leftrefill_amd import _lib, ops, packing  # noqa: E402


def main():
    lib = _lib.load()
    lib.lr_ffn_set_trace.argtypes = [ctypes.c_void_p]
    lib.lr_ffn_set_trace.restype = None
    d = torch.device("cuda:0")
    C, H, M = 320, 1280, 65536
    w1 = (torch.randn(2 * H, C) / C ** 0.5).half().to(d)
    b1 = torch.zeros(2 * H, device=d)
    w2 = packing.pack_pieces(torch.randn(C, H) / H ** 0.5).to(d)
    b2 = torch.zeros(C, device=d)
    xs = [torch.randn(M, C, device=d).half() for _ in range(8)]
    run = lambda x: ops.ffn_block(x, w1, b1, w2, b2, eps=1e-5)
    for x in xs[:4]:
        run(x)
    torch.cuda.synchronize()
    nb = M // 128
    trace = torch.zeros(nb * 8 * 16, device=d, dtype=torch.int64)
    lib.lr_ffn_set_trace(trace.data_ptr())
    run(xs[5])
    torch.cuda.synchronize()
    lib.lr_ffn_set_trace(None)
    tr = trace.reshape(nb, 8, 16).cpu().double()
    seg = lambda a, b_: (tr[:, :, b_] - tr[:, :, a]).mean().item()
    print(f"rows+LN {seg(0, 1):7.0f}  wait first piece {seg(1, 2):7.0f}")
    for j in range(3):
        a, am, b_, c_ = 2 + 4 * j, 3 + 4 * j, 4 + 4 * j, 5 + 4 * j
        nxt = 2 + 4 * (j + 1) if j < 2 else None
        print(f"  super-chunk {j}: A1 mfma+gelu {seg(a, am):6.0f}  wait {seg(am, b_):6.0f}  A2 + wait {seg(b_, c_):6.0f}" +
              (f"  C + wait {seg(c_, nxt):6.0f}" if nxt else ""))
    print(f"  main loop {seg(2, 14):8.0f}  ({seg(2, 14) / 60:6.0f} per step)   epilogue {seg(14, 15):7.0f}   block total {seg(0, 15):8.0f}")


if __name__ == "__main__":
    main()
