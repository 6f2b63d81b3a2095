"""Micro-benchmark of the one-launch GroupNorm (lr_groupnorm_apply_n on per-group partials) at the UNet's shapes, cold operands
(rotating over > 256 MB), inside a hipGraph:  [LR_GN_APPLY_BLOCKS=n] python tools/bench_gn_apply.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
tot_us = 0.0
for N, HW, C, chunks, count in ((8, 8192, 320, 32, 13), (8, 2048, 640, 8, 11), (8, 512, 1280, 4, 11), (8, 128, 1280, 4, 12), (8, 8192, 640, 32, 2),
                                (8, 2048, 1280, 8, 1)):
    per = N * HW * C * 2 * 2
    nsets = max(2, int(400e6 / per) + 1)
    xs = [torch.randn(N * HW, C, device=dev).half() for _ in range(nsets)]
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    gp = torch.rand(N, chunks, 32, 2, device=dev) * (HW * C / 32 / chunks)
    gp[..., 1] += gp[..., 0] ** 2 / (HW * C / 32 / chunks)
    ops.group_norm_groups(xs[0], N, HW, g, b, 1e-5, True, gp, chunks)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(2 * nsets):
            y = ops.group_norm_groups(xs[i % nsets], N, HW, g, b, 1e-5, True, gp, chunks)
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / (2 * nsets))
    tot_us += best * count
    print(f"gn apply N={N} HW={HW} C={C}: {best:6.1f} us  {per / best / 1e6:.2f} TB/s  (x{count} per step)")
print(f"sum over a UNet step (single-source GroupNorms): {tot_us:.0f} us   LR_GN_APPLY_BLOCKS={os.environ.get('LR_GN_APPLY_BLOCKS', '512 (default)')}")
