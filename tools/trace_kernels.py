"""Per-launch durations of the kernels whose name contains a substring, from a rocprofv3 --kernel-trace CSV, grouped by (name, grid):
    python tools/trace_kernels.py <kernel_trace.csv> <substring> [<marker substring for the step window>]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
    groups = []
    for i in idx:
        if not groups or i - groups[-1][-1] > 50:
            groups.append([i])
        else:
            groups[-1].append(i)
    rows = rows[groups[-2][-1] + 1:groups[-1][0]]
agg = defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][:48], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:48s} grid {g:>9s}  n={len(v):3d}  avg {sum(v) / len(v):8.1f} us  total {sum(v):8.1f} us")
