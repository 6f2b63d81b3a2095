"""Experiment (MI355X, developer variant built with tools/build_variant.sh stagger -DLR_GEMM_STAGGER): do two co-resident 4-wave GEMM
blocks of a CU overlap their memory and matrix phases if the second one starts late?

    LEFTREFILL_LIB_PATH=leftrefill_amd/lib/variants/libleftrefill_hip_stagger.so python tools/bench_stagger.py

Per short-K shape: the table's plan, then the 128 x {128, 160} 2-stage tiles (two blocks per CU) with LR_GEMM_STAGGER = 0 .. 24000 cycles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_shapes as bs  # noqa: E402
from leftrefill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for name, M, N, K, taps, fl in bs.SHAPES:
    if taps != 1 or fl.get("geglu"):
        continue
    n_out = N
    per_set = 2.0 * (M * K + M * n_out * (2 if fl.get("resid") else 1))
    nsets = max(2, int(600e6 / per_set) + 1)
    sets, launch = bs.make_case(M, N, K, taps, fl, dev, nsets)
    tm, tn, sp, stg = ops.gemm_plan(M, N, K, taps=taps, ln=bool(fl.get("ln")), stats=bool(fl.get("stats")))
    os.environ["LR_GEMM_STAGGER"] = "0"
    launch(sets[0], tm, tn, sp, stg)
    torch.cuda.synchronize()
    base = min(bs.time_seq(lambda i: launch(sets[i % nsets], tm, tn, sp, stg), nsets * 2) for _ in range(3))
    line = f"{name:24s} table {tm}x{tn}{'d' if stg else ''}: {base:6.1f} us |"
    for tile in ((128, 160), (128, 128)):
        if N % tile[1] and tile[1] == 160:
            continue
        line += f" {tile[0]}x{tile[1]}:"
        for cyc in (0, 2000, 4000, 8000, 12000, 16000, 24000):
            os.environ["LR_GEMM_STAGGER"] = str(cyc)
            launch(sets[0], tile[0], tile[1], 1, 0)
            torch.cuda.synchronize()
            t = min(bs.time_seq(lambda i: launch(sets[i % nsets], tile[0], tile[1], 1, 0), nsets * 2) for _ in range(3))
            line += f" {cyc // 1000}k {t:5.1f}"
    print(line, flush=True)
    del sets, launch
    torch.cuda.empty_cache()
