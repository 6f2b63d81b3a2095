"""Per-kernel breakdown of ONE steady-state step from a rocprofv3 --kernel-trace CSV.

    python tools/step_breakdown.py <kernel_trace.csv> <marker substring> [top]

The step is the window between the last two groups of kernels whose name contains the marker (e.g. `multi_tensor_apply` =
the optimizer of `bench.py --workload train`, `ddim_cfg_step` = the DDIM update of the sampling loop)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 50:
        groups.append([i])
    else:
        groups[-1].append(i)
a, b = groups[-2][-1], groups[-1][0]
seg = rows[a + 1:b]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
agg = defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:64]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ks = sum(v[1] for v in agg.values())
print(f"step span {span:.0f} us, kernel time {ks:.0f} us, {len(seg)} kernels")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:66s} n={v[0]:4d} {v[1]:8.0f} us {100 * v[1] / span:5.1f}%")
