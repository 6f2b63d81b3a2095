"""Micro-benchmark of the GroupNorm kernels inside a hipGraph (no host gaps): python tools/bench_gn.py N HW C [C2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

N, HW, C = [int(v) for v in sys.argv[1:4]]
dev = torch.device("cuda:0")
x = torch.randn(N * HW, C, device=dev).half()
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
ops.group_norm(x, N, HW, g, b, 1e-5, True)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        y = ops.group_norm(x, N, HW, g, b, 1e-5, True)
gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
gr.replay()
e1.record()
e1.synchronize()
us = 1e3 * e0.elapsed_time(e1) / 20
print(f"groupnorm N={N} HW={HW} C={C}: {us:.1f} us per stats+apply ({3 * N * HW * C * 2 / us / 1e6:.2f} TB/s of 3 passes)")
