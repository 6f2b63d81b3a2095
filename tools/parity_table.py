"""Whole-network parity table: every UNet / multi-view case of tests/test_gpu_unet.py, HIP path vs the
reference golden (or the oracle's fp32 forward where the case has no golden) and the oracle's fp16-autocast emulation vs the same
reference: rel-L2, max-abs, % of elements outside the north-star tolerance (rtol 2e-3 / atol 1e-3).  MI355X.

    python tools/parity_table.py [out.txt]        # default gpurun_out/parity_table.txt; copy to profiles/rNN_parity_table.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tests.test_gpu_unet as T  # noqa: E402
from oracle import golden_spec as G  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_table.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    gold = {n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")) for n in ("unet", "multiview")}
    rows = []
    for case, cname, N, H, W, ts in G.UNET_CASES:
        if cname == "SMALL":          # d_head 32: CPU-only config
            continue
        rows.append(("golden", T.measure_unet(case, cname, N, H, W, ts, ref=torch.from_numpy(gold["unet"][case]))))
    for case, V, concat, b, H, W in G.MV_CASES:
        n = b * (V - 1 if concat else V)
        rows.append(("golden", T.measure_unet(case, "MV", n, H, W, [501] * n, ref=torch.from_numpy(gold["multiview"][case]),
                                              multiview=(V, concat))))
    rows.append(("oracle fp32", T.measure_unet("unet_full_64x128_headline", "FULL", 2, 64, 128, [981, 21])))
    lines = ["# whole-network parity, HIP path (fp16) and the oracle's fp16-autocast emulation of the reference, both vs the fp32 reference",
             "# viol = % of output elements outside rtol 2e-3 / atol 1e-3 (north-star tolerance); produced by tools/parity_table.py on MI355X",
             f"{'case':28s} {'reference':12s} {'shape':18s} | {'HIP rel-L2':>10s} {'max-abs':>9s} {'viol %':>7s} | {'emu rel-L2':>10s} {'max-abs':>9s} "
             f"{'viol %':>7s} | {'HIP-vs-emu rel-L2':>17s} {'viol %':>7s}"]
    for kind, r in rows:
        lines.append(f"{r['case']:28s} {kind:12s} {str(r['shape']):18s} | {r['rel']:10.3e} {r['max_abs']:9.2e} {100 * r['viol']:7.3f} | "
                     f"{r['rel_emu']:10.3e} {r['max_abs_emu']:9.2e} {100 * r['viol_emu']:7.3f} | {r['rel_hip_vs_emu']:17.3e} "
                     f"{100 * r['viol_hip_vs_emu']:7.3f}")
    txt = "\n".join(lines) + "\n"
    open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
