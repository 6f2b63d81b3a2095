"""Per-shape micro-benchmark of lr_gemm_conv_f16 over the transformer / conv shapes of the configs[1] UNet step (MI355X).

    python tools/bench_shapes.py [--tiles all|table] [--only substr] [--reps 6]

Every launch of a timed sequence uses a different buffer set (rotating over > 256 MB, the Infinity Cache size), so the
"cold" column is what a kernel sees inside the UNet step; "hot" re-launches on one buffer set (what an autotuner that
times back-to-back launches would see).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import ops  # noqa: E402

TILES = list(ops.TILE_CANDIDATES)     # (tile_m, tile_n, pipe)
# name, M, N, K, taps, flags
SHAPES = [
    ("l0 KC resid+stats", 65536, 320, 320, 1, dict(resid=1, stats=1)),
    ("l0 KC ln (to_q)", 65536, 320, 320, 1, dict(ln=1)),
    ("l0 KC resid (proj_out)", 65536, 320, 320, 1, dict(resid=1)),
    ("l0 qkv ln", 65536, 960, 320, 1, dict(ln=1)),
    ("l0 geglu ln", 65536, 2560, 320, 1, dict(ln=1, geglu=1)),
    ("l0 ff2 resid+stats", 65536, 320, 1280, 1, dict(resid=1, stats=1)),
    ("l1 KC resid+stats", 16384, 640, 640, 1, dict(resid=1, stats=1)),
    ("l1 qkv ln", 16384, 1920, 640, 1, dict(ln=1)),
    ("l1 geglu ln", 16384, 5120, 640, 1, dict(ln=1, geglu=1)),
    ("l1 ff2 resid+stats", 16384, 640, 2560, 1, dict(resid=1, stats=1)),
    ("l2 KC resid+stats", 4096, 1280, 1280, 1, dict(resid=1, stats=1)),
    ("l2 qkv ln", 4096, 3840, 1280, 1, dict(ln=1)),
    ("l2 geglu ln", 4096, 10240, 1280, 1, dict(ln=1, geglu=1)),
    ("l2 ff2 resid+stats", 4096, 1280, 5120, 1, dict(resid=1, stats=1)),
    ("l0 conv3x3 emb", 65536, 320, 2880, 9, dict(rowvec=1)),
    ("l0 conv3x3 resid", 65536, 320, 2880, 9, dict(resid=1)),
    ("l1 conv3x3 resid", 16384, 640, 5760, 9, dict(resid=1)),
    ("l2 conv3x3 resid", 4096, 1280, 11520, 9, dict(resid=1)),
    ("l3 conv3x3 resid", 1024, 1280, 11520, 9, dict(resid=1)),
]


def make_case(M, N, K, taps, fl, dev, nsets):
    C = K // taps
    if taps == 9:
        B = 8
        HW = M // B
        W = int((HW * 2) ** 0.5)
        H = HW // W
    else:
        B, H, W = 1, 1, M
    n_out = N // 2 if fl.get("geglu") else N
    sets = []
    for _ in range(nsets):
        d = dict(x=torch.randn(M, C, device=dev).half(), out=torch.empty(M, n_out, device=dev, dtype=torch.float16))
        if fl.get("resid"):
            d["resid"] = torch.randn(M, n_out, device=dev).half()
        if fl.get("ln"):
            xf = d["x"].float()
            d["st"] = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).reshape(M, 1, 2).contiguous()
        sets.append(d)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    pm = os.environ.get("LR_BENCH_PM", "0") == "1"
    if pm:
        from leftrefill_amd import packing
        w = packing.pack_pm(w)
    b = torch.randn(N, device=dev)
    cs = (w.float().sum(2).sum(0) if pm else w.float().sum(1)).contiguous()
    rv = torch.randn(B, N, device=dev).half() if fl.get("rowvec") else None

    def launch(d, tm, tn, sp, stg=0):
        return ops.gemm_conv(d["x"], w, wt_pm=pm, B=B, H=H, W=W, taps=taps, bias=b, out=d["out"], tile_m=tm, tile_n=tn, splits=sp, pipe=stg,
                             geglu=bool(fl.get("geglu")), resid=d.get("resid"), rowvec=rv,
                             ln=(d["st"], 1e-5, cs) if fl.get("ln") else None, want_stats=bool(fl.get("stats")))
    return sets, launch


def time_seq(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="all")
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, M, N, K, taps, fl in SHAPES:
        if a.only and a.only not in name:
            continue
        n_out = N // 2 if fl.get("geglu") else N
        per_set = 2.0 * (M * (K // taps) + M * n_out * (2 if fl.get("resid") else 1))
        nsets = max(2, int(600e6 / per_set) + 1)
        sets, launch = make_case(M, N, K, taps, fl, dev, nsets)
        if a.tiles == "table":
            plans = [ops.gemm_plan(M, N, K, taps=taps, geglu=bool(fl.get("geglu")), ln=bool(fl.get("ln")),
                                   stats=bool(fl.get("stats")))]
        else:
            plans = [(tm, tn, 0, stg) for tm, tn, stg in TILES]
        res = []
        for tm, tn, sp, stg in plans:
            if fl.get("geglu") and tn == 160:
                continue
            if (fl.get("ln") or fl.get("stats")) and sp == 0:
                sp = 1
            try:
                launch(sets[0], tm, tn, sp, stg)
                torch.cuda.synchronize()
            except RuntimeError as e:
                res.append((f'{tm}x{tn}' + {0: '', 4: 'd', 8: 'h'}.get(stg, '?'), tn, None, None, str(e)[:40]))
                continue
            cold = min(time_seq(lambda i: launch(sets[i % nsets], tm, tn, sp, stg), nsets * 2) for _ in range(2))
            hot = min(time_seq(lambda i: launch(sets[0], tm, tn, sp, stg), a.reps) for _ in range(2))
            res.append((f'{tm}x{tn}' + {0: '', 4: 'd', 8: 'h'}.get(stg, '?'), tn, cold, hot, ""))
        fl_ = 2.0 * M * N * K
        ok = [r for r in res if r[2] is not None]
        best = min(ok, key=lambda r: r[2]) if ok else None
        line = "  ".join(f"{tm}:{c:6.1f}/{h:6.1f}" if c is not None else f"{tm}:  err" for tm, tn, c, h, _ in res)
        if best:
            print(f"{name:24s} M={M:6d} N={N:5d} K={K:5d} best {best[0]} {best[2]:7.1f} us cold "
                  f"({fl_ / best[2] / 1e6:6.0f} TF)  | {line}", flush=True)
        del sets, launch
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
