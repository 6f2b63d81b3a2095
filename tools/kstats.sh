#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run; prints per-kernel per-UNet-step totals.  usage: [BENCH_ARGS="--workload mv5 --mv-shard"] kstats.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
rm -rf /tmp/kt_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline $BENCH_ARGS > /tmp/b_$tag.json 2>/dev/null
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
cp $f $R/gpurun_out/kstats_$tag.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 54.0   # 50 timed + 4 preparation UNet steps in `bench.py --steps 1 --warmup 0`
tot = 0.0
for r in rows:
    ns = float(r["TotalDurationNs"])
    name = r["Name"]
    if "at::native" in name or "rocclr" in name or "rocblas" in name:
        continue
    tot += ns
    if ns / steps > 4e3:
        print(f"{name[:70]:70s} calls/step {int(r['Calls']) / steps:6.1f}  us/step {ns / steps / 1e3:8.1f}  avg us {float(r['AverageNs']) / 1e3:7.1f}")
print(f"total kernel time per step {tot / steps / 1e6:.3f} ms")
PY
python -c "
import json; r=json.load(open('/tmp/b_$tag.json')); print('$tag', r['value'], r['per_unet_step_ms'])"
