"""What the vendor GEMM (torch.nn.functional.linear -> hipBLASLt / rocBLAS) reaches on the UNet's short-K linear shapes,
next to lr_gemm_conv_f16 (autotuned tile).  Information only: the product path never calls the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for M, N, K in ((65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (65536, 960, 320), (65536, 2560, 320),
                (65536, 320, 1280), (16384, 5120, 640), (4096, 1280, 5120), (1024, 1280, 1280)):
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    bh = b.half()
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, bh))
    t_own = timeit(lambda: ops.gemm_conv(x, w, B=1, H=1, W=M, taps=1, bias=b))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: library {t_lib:7.1f} us ({fl / t_lib / 1e6:6.0f} TF)   lr_gemm_conv_f16 {t_own:7.1f} us "
          f"({fl / t_own / 1e6:6.0f} TF)")
