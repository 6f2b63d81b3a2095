"""Aggregate rocprofv3 --pmc counter CSVs per kernel family over the LAST eager UNet step of tools/pmc_step.py.

    python tools/pmc_util.py <dir_with_counter_collection_csv> [<dir2> ...]

Prints, per kernel family, the summed counters and the ratios used in DESIGN.md:
  MFMA busy    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   [rocprofv3 reports GRBM_GUI_ACTIVE summed
                 over the 8 XCDs; calibrated on the long-K convs: 1000-1200 TFLOP/s <-> 40-48 % busy]
  wave parked  = SQ_WAIT_ANY / SQ_WAVE_CYCLES,  issue stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def family(name):
    # (rocprofv3 reports mangled names for templates: ILi256ELi8ELi320ELi2ELi2E... = <256, 8, 320, 2, 2, ...>)
    for mangled, label in (("conv_halo_kernelILi320", "conv_halo_kernel<320,2,2> (halo-tile 3x3 conv, wave tile 8 lines x 80)"),
                           ("conv_halo_kernelILi160", "conv_halo_kernel<160,4,3> (halo-tile 3x3 conv, wave tile 4 lines x 80)"),
                           ("gemm_conv_pipe_kernelILi256ELi8ELi320ELi2ELi2ELi0", "gemm_conv_pipe_kernel<256,8,320,2,2> (wave tile 128x80)"),
                           ("gemm_conv_pipe_kernelILi256ELi8ELi320ELi4ELi2ELi1", "gemm_conv_pipe_kernel<256,8,320,4,2,GEGLU>"),
                           ("gemm_conv_pipe_kernelILi256ELi8ELi256", "gemm_conv_pipe_kernel<256,8,256,2,2>"),
                           ("gemm_conv_pipe_kernelILi256ELi8ELi160", "gemm_conv_pipe_kernel<256,8,160,4,3>"),
                           ("gemm_conv_pipe_kernelILi256ELi8ELi128", "gemm_conv_pipe_kernel<256,8,128,4,3>"),
                           ("gemm_conv_pipe_kernelILi128", "gemm_conv_pipe_kernel<128,4,*,2,4> (4-stage)")):
        if mangled in name:
            return label
    for key in ("gemm_conv_pipe_kernel", "gemm_conv_kernel", "xattn_block_kernel", "xattn640_kernel", "stin_block_kernel", "rowlin_kernel", "ffn_block_kernel", "gn_apply", "gn_stats", "gn_finalize", "layernorm",
                "splitk_reduce", "transpose_v"):
        if key in name:
            return key
    if "attention_kernel" in name:      # attention_kernel<T, VM, CAUSAL, EXACT>: VM 0 = register transpose, 1 = pre-transposed V^T, 2 = LDS transpose read
        for vm, label in (("Li1E", "pre-transposed V^T by LDS-DMA"), ("Li2E", "natural V, LDS transpose read"), ("Li0E", "V transposed in registers")):
            if "attention_kernelIDF16_" + vm in name or "attention_kernelIDF16b" + vm in name:
                return f"attention_kernel ({label})"
        return "attention_kernel"
    return None


tot = defaultdict(lambda: defaultdict(float))
for d in sys.argv[1:]:
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # last step = everything after the second-to-last timestep_embedding launch's dispatch
    marks = sorted({int(r["Dispatch_Id"]) for r in rows if "timestep_embedding" in r["Kernel_Name"]})
    start = marks[-1]
    for r in rows:
        if int(r["Dispatch_Id"]) < start:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
            tot[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            tot[fam]["launches:" + r["Counter_Name"]] += 1
for fam, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    out = [f"{fam:34s}"]
    if c.get("GRBM_GUI_ACTIVE"):
        out.append(f"MFMA busy {100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 256 * 4):5.1f}%")
    if c.get("SQ_WAVE_CYCLES"):
        out.append(f"parked {100 * c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:5.1f}%  issue-stall "
                   f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:5.1f}%  active {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:5.1f}%")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"LDS conflict {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:5.2f}%")
    print("  ".join(out))
