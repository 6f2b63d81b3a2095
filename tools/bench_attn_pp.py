"""A/B of the two attention forward kernels on the UNet's self-attention shapes (MI355X only), interleaved rounds in one process:

    python tools/bench_attn_pp.py [rounds]

LR_ATTN_PP modes: 0 = attention_kernel (4-wave blocks, two per CU), 2 = ping-pong kernel with 512-query blocks, 3 = with 256-query
blocks, 1 = the library's own choice.  Times include nothing but the attention launch (V^T is prepared once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

SHAPES = [("l0 self 8192^2", 8, 5, 8192, 8192), ("l1 self 2048^2", 8, 10, 2048, 2048), ("l2 self 512^2", 8, 20, 512, 512),
          ("mv5 l0 20480^2", 2, 5, 20480, 20480), ("mv5 l1 5120^2", 2, 10, 5120, 5120), ("mv4 l0 16384^2", 2, 5, 16384, 16384)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
for name, B, heads, Nq, Nkv in SHAPES:
    C = heads * 64
    sets = []
    for _ in range(3):
        qkv = torch.randn(B * Nq, 3 * C, device=dev).half()
        vt = ops.transpose_v(qkv[:, 2 * C:], B, heads, Nkv)
        sets.append((qkv, vt, torch.empty(B * Nq, C, device=dev, dtype=torch.float16)))
    res = {}
    for r in range(rounds):
        for mode in os.environ.get("LR_BENCH_MODES", "0,2,3,1").split(","):
            os.environ["LR_ATTN_PP"] = mode
            f = lambda i: ops.attention(sets[i % 3][0][:, :C], sets[i % 3][0][:, C:2 * C], sets[i % 3][0][:, 2 * C:], B, heads, Nq, Nkv,
                                        0.125, out=sets[i % 3][2], vt=sets[i % 3][1])
            f(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(6):
                f(i)
            e1.record()
            e1.synchronize()
            res.setdefault(mode, []).append(1e3 * e0.elapsed_time(e1) / 6)
    fl = 4.0 * B * heads * Nq * Nkv * 64
    line = "  ".join(f"mode {m}: {min(v):7.1f} us (median {sorted(v)[len(v) // 2]:7.1f}) {fl / min(v) / 1e6:6.0f} TF" for m, v in res.items())
    print(f"{name:18s} {line}", flush=True)
os.environ.pop("LR_ATTN_PP", None)
