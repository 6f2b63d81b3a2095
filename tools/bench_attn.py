"""Micro-benchmark of lr_attention_f16 (MI355X only): python tools/bench_attn.py B heads Nq Nkv [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

B, heads, Nq, Nkv = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
C = heads * 64
dev = torch.device("cuda:0")
qkv = torch.randn(B * Nq, 3 * C, device=dev).half()
kv = qkv if Nq == Nkv else torch.randn(B * Nkv, 3 * C, device=dev).half()
f = lambda: ops.attention(qkv[:, :C], kv[:, C:2 * C], kv[:, 2 * C:], B, heads, Nq, Nkv, 0.125)
if len(sys.argv) > 6:
    ops.VT_MIN_KEYS = int(sys.argv[6])      # 1: always pre-transpose V; huge: never
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f()
e1.record()
e1.synchronize()
us = 1e3 * e0.elapsed_time(e1) / reps
print(f"attention B={B} h={heads} {Nq}x{Nkv}: {us:.1f} us  {4.0 * B * heads * Nq * Nkv * 64 / us / 1e6:.1f} TFLOP/s")
