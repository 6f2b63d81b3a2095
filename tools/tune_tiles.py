"""Developer tool (MI355X only): time every (tile, split-K) plan for every GEMM shape of the shipped workloads and write
the table that `leftrefill_amd/tile_table.json` ships in-tree.

    LEFTREFILL_AUTOTUNE=1 python tools/tune_tiles.py [--out gpurun_out/tile_table.json] [--workloads single,cfg0,mv5,mv5shard,train,vae]

The product never times anything: ops.gemm_conv looks the plan up in the committed table (a pure function of the shape).
"""
import argparse
import json
import os
import sys

os.environ["LEFTREFILL_AUTOTUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from leftrefill_amd import ops  # noqa: E402


def unet_pass(model, B, h, w, device):
    unet = model.model.diffusion_model
    batch = bench.synthetic_batch(B, h, w, device, 7)
    c_concat, c_cross, uc_cross, x_T = batch
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
    t = torch.full((2 * B,), 501, device=device, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half()
    unet.use_hip_graph = False
    with torch.no_grad():
        unet(x, t, ctx)
    torch.cuda.synchronize()
    unet.use_hip_graph = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_table.json"))
    ap.add_argument("--workloads", default="single,cfg0,mv5,train,vae")
    ap.add_argument("--fresh", action="store_true", help="ignore the committed table (re-tune every shape)")
    ap.add_argument("--retune-conv", action="store_true",
                    help="re-time only the 3x3 stride-1 entries (all tile candidates incl. the halo-tile instances), keep the rest of the table")
    ap.add_argument("--retune-pointwise", action="store_true",
                    help="re-time only the taps = 1 entries (the grouped tile order of round 5 changed what multi-round pointwise plans cost), keep the rest")
    a = ap.parse_args()
    assert ops.AUTOTUNE
    if a.fresh:
        ops.tile_cache().clear()
    if a.retune_conv:
        for k in [k for k in ops.tile_cache() if k.split(",")[3:6] == ["9", "1", "0"]]:
            del ops.tile_cache()[k]
    if a.retune_pointwise:
        for k in [k for k in ops.tile_cache() if k.split(",")[3] == "1"]:
            del ops.tile_cache()[k]
    device = torch.device("cuda:0")
    wl = a.workloads.split(",")
    model = None
    if "single" in wl or "cfg0" in wl or "train" in wl:
        model = bench.build_model(device, "single")
    if "single" in wl:
        unet_pass(model, 4, 64, 128, device)
        print("single:", len(ops.tile_cache()), "shapes", flush=True)
    if "cfg0" in wl:
        unet_pass(model, 1, 32, 64, device)
        print("cfg0:", len(ops.tile_cache()), "shapes", flush=True)
    if "train" in wl:
        class A:
            steps, warmup, recompute, train_graph = 1, 1, False, False
        bench.train_bench(A, 0, 1, device, model=model, steps=1)
        print("train:", len(ops.tile_cache()), "shapes", flush=True)
    if "mv5" in wl:
        del model
        torch.cuda.empty_cache()
        mv = bench.build_model(device, "mv5")
        unet_pass(mv, 4, 64, 128, device)
        del mv
        print("mv5:", len(ops.tile_cache()), "shapes", flush=True)
    if "mv5shard" in wl:      # one rank of the 4-rank canvas-sharded multi-view job (peers simulated): UNet batch 2, own-row attention shapes
        torch.cuda.empty_cache()
        os.environ["LEFTREFILL_MV_SIM_WORLD"] = "4"
        mv = bench.build_model(device, "mv5")
        mv.model.diffusion_model.mv_shard = True
        for split in ("1", "0"):
            from leftrefill_amd import engine
            engine.MV_SPLIT_TARGET = split == "1"
            unet_pass(mv, 1, 64, 128, device)
        engine.MV_SPLIT_TARGET = True
        del mv
        os.environ.pop("LEFTREFILL_MV_SIM_WORLD")
        print("mv5shard:", len(ops.tile_cache()), "shapes", flush=True)
    if "vae" in wl:
        torch.cuda.empty_cache()
        bench.vae_timing(4, device)
        print("vae:", len(ops.tile_cache()), "shapes", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({k: list(v) for k, v in sorted(ops.tile_cache().items())}, f, indent=0)
    print("wrote", a.out, len(ops.tile_cache()), "entries")


if __name__ == "__main__":
    main()
