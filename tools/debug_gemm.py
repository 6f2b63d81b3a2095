import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leftrefill_amd import ops
d = torch.device("cuda:0")
M, N, K = 128, 64, 64
x = torch.randn(M, K, device=d).half()
w = torch.randn(N, K, device=d).half()
for tm, tn in ((128, 64), (128, 128), (256, 128)):
    y = ops.gemm_conv(x, w, B=1, H=1, W=M, taps=1, tile_m=tm, tile_n=tn)
    ref = x.float() @ w.float().t()
    err = (y.float() - ref).abs()
    bad = err > 0.05
    print(tm, tn, "bad frac", bad.float().mean().item())
    # pattern by (m%16, n%16 // 4), and which n-tile
    bm = bad.reshape(M // 16, 16, N // 16, 4, 4).float()
    print(" by ntile:", bm.mean(dim=(0, 1, 3, 4)).tolist())
    print(" by n quad (fq):", bm.mean(dim=(0, 1, 2, 4)).tolist())
    print(" by mtile:", bm.mean(dim=(1, 2, 3, 4)).tolist())
    # find where values actually come from: for a wrong element find matching ref element
    if bad.any():
        idx = bad.nonzero()[0]
        m, n = idx.tolist()
        v = y[m, n].float()
        cand = ((ref - v).abs() < 2e-2).nonzero()
        print(" y[%d,%d]=%.3f ref=%.3f candidates:" % (m, n, v, ref[m, n]), cand[:6].tolist())
