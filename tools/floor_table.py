"""Speed-of-light table per kernel family from a bench.py JSON line, and the stop rule that goes with it.

    python tools/floor_table.py BENCH.json [--fixed profiles/r06_fixed_cost.json] [--out profiles/r06_floor_table.txt]

For every `kernel_table` row of the instrumented UNet step:
    floor = max( FLOPs / sustained matrix peak   (roofline.sustained_peak_measured of the same run: random-operand MFMA chains, no memory),
                 algorithmic bytes / 6.3 TB/s    (every operand read once, every output written once; the HBM rate stream kernels reach),
                 fixed cost                      (launch + pipeline fill + epilogue of the same (M, N) with ONE K-step, tools/fixed_cost.py) )
sorted by (measured - floor) x launches per step: where the step's time is still above what the chip could do.
Stop rule: a family within 1.25x of its floor is CLOSED -- no more variants on it.
"""
import argparse
import json
import re
import sys

HBM_TBPS = 6.3


def load_line(path):
    for line in open(path):
        if line.startswith("{") and '"kernel_table"' in line:
            return json.loads(line)
    raise SystemExit(f"{path}: no bench JSON line with a kernel_table")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench")
    ap.add_argument("--fixed", default=None, help="JSON of tools/fixed_cost.py: shape -> us of the 1-K-step launch")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = load_line(a.bench)
    rf = r["roofline"]
    peak = rf.get("sustained_peak_measured") or 1800.0
    fixed = json.load(open(a.fixed)) if a.fixed else {}
    rows = []
    for t in r["kernel_table"]:
        if "gflop" not in t:
            raise SystemExit("bench JSON predates the gflop / alg_mb columns: re-run bench.py")
        f_us = t["gflop"] / peak * 1e3                       # GFLOP / (TFLOP/s) = ms; -> us
        b_us = t["alg_mb"] / (HBM_TBPS * 1e6) * 1e6          # MB / (TB/s) = us
        x_us = float(fixed.get(t["shape"], 0.0))
        floor = max(f_us, b_us, x_us)
        which = "matrix" if floor == f_us else "bytes" if floor == b_us else "fixed"
        rows.append(dict(t, floor_us=floor, bound=which, flop_us=f_us, byte_us=b_us, fixed_us=x_us, ratio=t["avg_us"] / floor if floor > 0 else float("inf"),
                         gap_us=(t["avg_us"] - floor) * t["n"]))
    rows.sort(key=lambda x: -x["gap_us"])
    tot = sum(x["total_us"] for x in rows)
    tot_floor = sum(x["floor_us"] * x["n"] for x in rows)
    lines = []
    w = lines.append
    w(f"Floor table of the instrumented UNet step (bench.py kernel_table; eager launches with HIP events).  per_unet_step_ms {r['per_unet_step_ms']:.3f}, "
      f"sclk {rf.get('sclk_mhz_mean', 0):.0f} MHz, {rf.get('power_w_mean', 0):.0f} W")
    w(f"floor = max(FLOPs / {peak:.0f} TFLOP/s sustained matrix peak of this run, algorithmic bytes / {HBM_TBPS} TB/s, fixed 1-K-step cost"
      + (f" from {a.fixed}" if a.fixed else " (not measured: 0)") + ")")
    w(f"rows: {len(rows)}; measured {tot / 1e3:.2f} ms, floors {tot_floor / 1e3:.2f} ms ({tot / tot_floor:.2f}x); CLOSED = within 1.25x of the floor (stop rule: no more variants)")
    w("")
    w(f"{'shape':78s} {'n':>3s} {'avg us':>8s} {'floor':>7s} {'bound':>6s} {'x':>5s} {'gap us/step':>11s}  {'matrix':>7s} {'bytes':>7s} {'fixed':>6s}  status")
    for x in rows:
        st = "CLOSED" if x["ratio"] <= 1.25 else ("open" if x["gap_us"] >= 50 else "open (small)")
        w(f"{x['shape'][:78]:78s} {x['n']:3d} {x['avg_us']:8.1f} {x['floor_us']:7.1f} {x['bound']:>6s} {x['ratio']:5.2f} {x['gap_us']:11.1f}  "
          f"{x['flop_us']:7.1f} {x['byte_us']:7.1f} {x['fixed_us']:6.1f}  {st}")
    closed = sum(x["total_us"] for x in rows if x["ratio"] <= 1.25)
    w("")
    w(f"closed families carry {closed / 1e3:.2f} of {tot / 1e3:.2f} ms; the open ones lose {sum(x['gap_us'] for x in rows if x['ratio'] > 1.25) / 1e3:.2f} ms per step to their floors")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
