"""lr_attention_f16 with 256-query blocks (two 32-query blocks per wave, two blocks per CU) vs 128-query blocks (one per wave, three
blocks per CU) on the attention shapes of the UNet step (MI355X only), interleaved rounds in one process:

    python tools/bench_attn_nqb.py [rounds]

LR_ATTN_NQB = 2 | 1 forces the block size, unset = the library's rule (launch_attention).  The two must agree bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

SHAPES = [("l0 self 8192^2", 8, 5, 8192, 8192), ("l1 self 2048^2", 8, 10, 2048, 2048), ("l2 self 512^2", 8, 20, 512, 512),
          ("l3 self 128^2", 8, 20, 128, 128), ("l1 cross 2048x77", 8, 10, 2048, 77), ("l2 cross 512x77", 8, 20, 512, 77),
          ("l3 cross 128x77", 8, 20, 128, 77), ("mv5 l0 20480^2", 2, 5, 20480, 20480), ("mv5 l1 5120^2", 2, 10, 5120, 5120),
          ("cfg0 l0 2048^2 B2", 2, 5, 2048, 2048), ("cfg0 l1 512^2 B2", 2, 10, 512, 512)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")


def timed(f, n=6):
    f(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for name, B, heads, Nq, Nkv in SHAPES:
    C = heads * 64
    sets = []
    for _ in range(3):
        q = torch.randn(B * Nq, C, device=dev).half()
        kv = torch.randn(B * Nkv, 2 * C, device=dev).half()
        sets.append((q, kv, torch.empty(B * Nq, C, device=dev, dtype=torch.float16)))
    outs, res = {}, {}
    for r in range(rounds):
        for mode in ("2", "1", "auto"):
            if mode == "auto":
                os.environ.pop("LR_ATTN_NQB", None)
            else:
                os.environ["LR_ATTN_NQB"] = mode
            f = lambda i: ops.attention(sets[i % 3][0], sets[i % 3][1][:, :C], sets[i % 3][1][:, C:], B, heads, Nq, Nkv, 0.125, out=sets[i % 3][2])
            res.setdefault(mode, []).append(timed(f))
            if r == 0:
                outs[mode] = f(0).clone()
    os.environ.pop("LR_ATTN_NQB", None)
    same = all(torch.equal(outs["2"], o) for o in outs.values())
    fl = 4.0 * B * heads * Nq * Nkv * 64
    line = "  ".join(f"nqb {m}: {min(v):7.1f} us {fl / min(v) / 1e6:5.0f} TF" for m, v in res.items())
    print(f"{name:20s} {line}  bit-identical: {same}", flush=True)
