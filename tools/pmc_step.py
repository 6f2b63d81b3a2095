"""HBM traffic of the GEMM family over ONE eager UNet step (configs[1], UNet batch 8), from rocprofv3 PMC counters.

Two roles:

  python tools/pmc_step.py run
      the workload to put under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domains):
      builds the bench model, tunes tiles on a first eager step, then runs ONE more eager step (no hipGraph -- the
      counter collection does not survive graph replay on this stack).

  python tools/pmc_step.py reduce <fetch_dir> <write_dir> <out.json>
      reads the two counter_collection CSVs, keeps the LAST step's gemm_conv* dispatches (one per ops.gemm_conv call,
      their number is `launches_per_unet_step` of bench.py) and writes per-launch averages, applying the gfx950
      correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B -> x2; both counters are in KiB).
"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build_model(dev)
    B = 4
    c_concat, c_cross, uc_cross, x_T = bench.synthetic_batch(B, 64, 128, dev, 1234)
    unet = model.model.diffusion_model
    unet.use_hip_graph = False
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
    t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half()
    from leftrefill_amd import ops
    descs = []
    orig = ops.gemm_conv

    def spy(*a, **k):
        descs.append(dict(M=k["B"] * k["H"] * k["W"], N=a[1].shape[0], K=a[1].shape[1], taps=k.get("taps", 1),
                          stride=k.get("stride", 1), up=k.get("up", 0), geglu=bool(k.get("geglu", False)),
                          resid=k.get("resid") is not None))
        return orig(*a, **k)

    def arm():
        ops.gemm_conv = spy
    with torch.no_grad():
        for i in range(2):      # the launches of the captured step (context K / V projections cached, see bench.eager_unet_step)
            bench.eager_unet_step(unet, x, t, ctx, hook=arm if i == 1 else None)
            torch.cuda.synchronize()
    ops.gemm_conv = orig
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(descs, open("gpurun_out/pmc_descs.json", "w"))
    print("pmc_step: done", len(descs))


def _rows(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    out = []
    for f in files:
        out += list(csv.DictReader(open(f)))
    out.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    return out


FAMILY = ("gemm_conv", "conv_halo_kernel")      # the GEMM family of bench.py's roofline object (kernel-name substrings)


def _family_groups(rows, counter):
    """Counter value (bytes) per ops.gemm_conv CALL, in dispatch order: a family kernel plus the splitk_reduce launch that follows it.
    (Until the middle of round 5 this selected by "gemm_conv" alone: the conv_halo launches fell out, their descriptors did not, and the
    per-launch averages were taken over a set that reached back into the previous step -- see profiles/README.md, round-5 erratum.)"""
    out = []
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]) * 1024.0
        name = r["Kernel_Name"]
        if any(k in name for k in FAMILY) and "splitk_reduce" not in name:
            out.append(v)
        elif "splitk_reduce" in name and out:
            out[-1] += v
    return out


def reduce_(fetch_dir, write_dir, out_path, launches=194):
    res = {}
    per = {}
    for key, d, counter, corr in (("fetch", fetch_dir, "FETCH_SIZE", 2.0), ("write", write_dir, "WRITE_SIZE", 1.0)):
        groups = _family_groups(_rows(d), counter)
        # pmc_step.run makes two eager steps; the context K / V projections (gemm_conv calls outside the armed region) come on top, so the
        # count is bounded, not exact -- the instrumented step is the TAIL of the process, and the per-shape check below (no kernel writes
        # less than its output) is what proves the alignment
        if not 2 * launches <= len(groups) <= 2 * launches + 64:
            raise RuntimeError(f"{counter}: {len(groups)} GEMM-family calls in the trace for {launches} per step -- refusing to average a misaligned set")
        last = groups[-launches:]
        res[key + "_bytes_per_launch"] = corr * sum(last) / len(last)
        res[key + "_launches"] = len(last)
        per[key] = [corr * v for v in last]
    if os.path.exists("gpurun_out/pmc_descs.json"):
        descs = json.load(open("gpurun_out/pmc_descs.json"))
        assert len(descs) == len(per["fetch"]), (len(descs), len(per["fetch"]))
        if True:
            agg = {}
            for d, f, w in zip(descs, per["fetch"], per["write"]):
                src_rows = d["M"] * (4 if d["stride"] == 2 else 1) / (4 if d["up"] else 1)
                n_out = d["N"] // 2 if d["geglu"] else d["N"]
                alg_r = 2.0 * (src_rows * d["K"] / d["taps"] + d["N"] * d["K"] + (d["M"] * n_out if d["resid"] else 0))
                alg_w = 2.0 * d["M"] * n_out
                k = f'{d["M"]}x{d["N"]}x{d["K"]} t{d["taps"]} s{d["stride"]} u{d["up"]}' + (" geglu" if d["geglu"] else "")
                a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += f; a[2] += alg_r; a[3] += w; a[4] += alg_w
            rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
            res["per_shape"] = [dict(shape=k, n=a[0], fetch_mb=a[1] / a[0] / 1e6, alg_read_mb=a[2] / a[0] / 1e6,
                                     write_mb=a[3] / a[0] / 1e6, alg_write_mb=a[4] / a[0] / 1e6) for k, a in rows]
            bad = [r for r in res["per_shape"] if r["write_mb"] < 0.9 * r["alg_write_mb"]]
            if bad:      # a kernel cannot write less than its output: the alignment is wrong again
                raise RuntimeError(f"per-shape write bytes below the output size: {bad[:3]}")
            for r in res["per_shape"][:30]:
                print(f'{r["shape"]:38s} n={r["n"]:2d} fetch {r["fetch_mb"]:8.1f} MB (alg {r["alg_read_mb"]:7.1f})  '
                      f'write {r["write_mb"]:7.1f} (alg {r["alg_write_mb"]:7.1f})')
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    res["note"] = ("GEMM family, last eager UNet step at batch 8 (configs[1]); FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE "
                   "as reported (uncalibrated per the guide); separate --pmc passes")
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        reduce_(sys.argv[2], sys.argv[3], sys.argv[4], *(int(v) for v in sys.argv[5:6]))
