"""Average rocprofv3 --pmc counters per dispatch for kernels whose name contains a substring.

    rocprofv3 --pmc A B C -d out1 -- <cmd> ;  rocprofv3 --pmc D E -d out2 -- <cmd>
    python tools/pmc_kernel.py <substring> out1 [out2 ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

sub = sys.argv[1]
tot, cnt = defaultdict(float), defaultdict(int)
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
avg = {k: tot[k] / cnt[k] for k in tot}
for k in sorted(avg):
    print(f"{k:32s} {avg[k]:16.0f}   ({cnt[k]} dispatches)")
wc = avg.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
              "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
        if k in avg:
            print(f"  {k:28s} / SQ_WAVE_CYCLES = {100 * avg[k] / wc:5.1f} %")
if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    print(f"  MFMA busy = {100 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / 8 * 256 * 4):5.1f} % of SIMD-cycles")
if "SQ_INSTS_VALU" in avg and "SQ_INSTS_MFMA" in avg:
    print(f"  VALU (incl. MFMA) per MFMA = {avg['SQ_INSTS_VALU'] / avg['SQ_INSTS_MFMA']:.2f}")
