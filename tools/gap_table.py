"""Launch-to-launch gaps from a rocprofv3 --kernel-trace CSV:  python tools/gap_table.py <kernel_trace.csv> [substring ...]

For every kernel whose name contains one of the substrings (default: the 256-tile GEMM-family instances), the gap between its end and the
start of the next kernel on the same queue, plus its own duration: median / mean over the trace.

Caveat (round 5): in a hipGraph replay rocprofv3 reports End(i) == Start(i + 1) for dependent kernels -- every gap reads 0.00 us and whatever
the boundary costs is inside the neighbours' durations.  A zero here is NOT "no boundary cost"; use it on eager traces only."""
import csv
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:] or ["conv_halo_kernelILi320", "conv_halo_kernelILi160", "gemm_conv_pipe_kernelILi256ELi8ELi320", "attention_kernel", "gn_apply"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps, durs = defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if a.get("Queue_Id") != b.get("Queue_Id"):
        continue
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 200000:      # (a host-side pause between samplings, not a boundary)
        continue
    for k in keys:
        if k in a["Kernel_Name"]:
            gaps[k].append(g / 1e3)
            durs[k].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
for k in keys:
    if gaps[k]:
        print(f"{k:44s} n={len(gaps[k]):5d}  gap after: median {statistics.median(gaps[k]):6.2f} us  mean {statistics.mean(gaps[k]):6.2f} us"
              f"   own duration: median {statistics.median(durs[k]):7.1f} us")
