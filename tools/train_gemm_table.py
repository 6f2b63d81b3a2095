"""Per-shape table of the GEMM-family launches of ONE eager training step (bench.py --workload train): HIP events around every
ops.gemm_conv launch (forward and input-gradient GEMMs), aggregated by shape.

    python tools/train_gemm_table.py [--dtype bf16] [--top 60]
"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from leftrefill_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
a.steps, a.warmup, a.train_graph, a.recompute, a.task = 2, 2, False, False, "nvs"
rec = []
orig = ops.gemm_conv
armed = [False]


def wrapped(*args, **k):
    if not armed[0]:
        return orig(*args, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(*args, **k)
    e1.record()
    rec.append((e0, e1, dict(M=k["B"] * k["H"] * k["W"], N=args[1].shape[-2], K=args[1].shape[-1], taps=k.get("taps", 1), stride=k.get("stride", 1),
                             up=k.get("up", 0), geglu=bool(k.get("geglu")), cat=k.get("x2") is not None, resid=k.get("resid") is not None)))
    return out


ops.gemm_conv = wrapped
armed[0] = True
res = bench.train_bench(a, 0, 1, torch.device("cuda:0"))
armed[0] = False
torch.cuda.synchronize()
steps = a.steps + max(1, a.warmup)
agg = defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, d in rec:
    key = f'{d["M"]}x{d["N"]}x{d["K"]} taps{d["taps"]} s{d["stride"]} up{d["up"]}' + (" geglu" if d["geglu"] else "") + (" cat" if d["cat"] else "") + (" resid" if d["resid"] else "")
    v = agg[key]
    v[0] += 1
    v[1] += 1e3 * e0.elapsed_time(e1)
    v[2] = 2.0 * d["M"] * d["N"] * d["K"]
tot = sum(v[1] for v in agg.values()) / steps
print(f"eager training step ({a.dtype}): {res['ms_per_step']:.2f} ms per step; GEMM-family launches {len(rec) / steps:.0f} per step, {tot / 1e3:.2f} ms per step")
print(f"{'shape (M x N x K)':58s} {'n/step':>6s} {'avg us':>8s} {'us/step':>8s} {'TFLOP/s':>8s}")
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{key:58s} {v[0] / steps:6.1f} {v[1] / v[0]:8.1f} {v[1] / steps:8.1f} {v[2] / (v[1] / v[0] * 1e-6) / 1e12:8.0f}")
