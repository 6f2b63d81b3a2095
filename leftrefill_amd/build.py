"""Build the gfx950 shared library (C ABI of include/leftrefill_hip.h) in-tree with hipcc.

    python -m leftrefill_amd.build            # -> leftrefill_amd/lib/libleftrefill_hip.so

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libleftrefill_hip.so")
SOURCES = ["norm.hip", "elementwise.hip", "gemm_conv.hip", "conv_halo.hip", "attention.hip", "attention_bwd.hip", "xattn_block.hip", "xattn_block640.hip", "stin_block.hip", "rowlin.hip", "ffn_block.hip", "conv_out.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# attention: keep MFMA accumulators in VGPRs -- the softmax VALU stream reads every S^T value and rescales O, and
# AGPR accumulators cost ~150 v_accvgpr_read/write per 64-key tile.
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "xattn_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "ffn_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "stin_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "rowlin.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attention_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "chain_common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(HERE, "..", "include", "leftrefill_hip.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append([cc] + FLAGS + EXTRA.get(s, []) + ["-c", src, "-o", obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
