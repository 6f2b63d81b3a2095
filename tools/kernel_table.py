"""Per-shape table from `bench.py --dump-kernels FILE` (one JSON record per launch of an eager UNet step).

    python tools/kernel_table.py FILE [rows]
"""
import json
import sys
from collections import defaultdict

rows = [json.loads(l) for l in open(sys.argv[1])]
agg = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = r["kernel"]
    if k == "gemm_conv":
        key = ("gemm", r["M"], r["N"], r["K"], r["taps"], r["stride"], r["up"], r["geglu"], r["cat"])
    elif k == "attention":
        key = ("attn", r["B"], r["heads"], r["Nq"], r["Nkv"])
    elif k == "xattn_block":
        key = ("xattn_block", r["M"], r["C"], r["Lc"], "pre" if r.get("pre") else "")
    elif k == "ffn_block":
        key = ("ffn_block", r["M"], r["C"], r["H"], "post" if r.get("post") else "")
    else:
        key = (k,)
    a = agg[key]
    a[0] += 1
    a[1] += r["us"]
    a[2] = r["tflops"]
tot = sum(a[1] for a in agg.values())
by = defaultdict(float)
for k, a in agg.items():
    by[k[0]] += a[1]
print(f"total {tot:.0f} us  " + "  ".join(f"{k} {v:.0f} us" for k, v in sorted(by.items(), key=lambda kv: -kv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{str(k):62s} n={a[0]:3d} tot={a[1]:8.1f}us ({100 * a[1] / tot:4.1f}%) avg={a[1] / a[0]:7.1f}us {a[2]:6.0f} TF")
