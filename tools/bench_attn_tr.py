"""How V reaches the PV product of the attention forward (MI355X only), interleaved rounds in one process:

    python tools/bench_attn_tr.py [rounds]

  vt   lr_transpose_v_f16 (timed separately) + lr_attention_vt_f16: pre-transposed, key-permuted V^T streams by LDS-DMA, one ds_read_b128 per fragment
  tr   lr_attention_f16 on the natural V: LDS-DMA + two ds_read_b64_tr_b16 per fragment (no copy of V)
  reg  lr_attention_f16 with LR_ATTN_TR=0: V transposed in registers on its way to LDS
The three must agree bit for bit (same MFMA operands in the same order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

SHAPES = [("l0 self 8192^2", 8, 5, 8192, 8192), ("l1 self 2048^2", 8, 10, 2048, 2048), ("l2 self 512^2", 8, 20, 512, 512),
          ("l3 self 128^2", 8, 20, 128, 128), ("l1 cross 2048x77", 8, 10, 2048, 77), ("l2 cross 512x77", 8, 20, 512, 77),
          ("l3 cross 128x77", 8, 20, 128, 77), ("mv5 l0 20480^2", 2, 5, 20480, 20480)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
ops.VT_MIN_KEYS = 1 << 30          # the tool chooses the path itself


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
        sets.append((q, kv, ops.transpose_v(kv[:, C:], B, heads, Nkv), torch.empty(B * Nq, C, device=dev, dtype=torch.float16)))
    outs = {}
    res = {}
    for r in range(rounds):
        for mode in ("vt", "tr", "reg"):
            os.environ["LR_ATTN_TR"] = "0" if mode == "reg" else "1"
            f = lambda i, m=mode: ops.attention(sets[i % 3][0], sets[i % 3][1][:, :C], sets[i % 3][1][:, C:], B, heads, Nq, Nkv, 0.125,
                                                out=sets[i % 3][3], vt=sets[i % 3][2] if m == "vt" else None)
            res.setdefault(mode, []).append(timed(f))
            if r == 0:
                outs[mode] = f(0).clone()
        res.setdefault("transpose_v", []).append(timed(lambda i: ops.transpose_v(sets[i % 3][1][:, C:], B, heads, Nkv, out=sets[i % 3][2])))
    os.environ.pop("LR_ATTN_TR", None)
    same = all(torch.equal(outs["vt"], o) for o in outs.values())
    fl = 4.0 * B * heads * Nq * Nkv * 64
    line = "  ".join(f"{m}: {min(v):7.1f} us" + (f" {fl / min(v) / 1e6:5.0f} TF" if m != "transpose_v" else "") for m, v in res.items())
    print(f"{name:20s} {line}  bit-identical: {same}", flush=True)
