"""Developer tool (MI355X only): choose GEMM plans by what they cost INSIDE the UNet step instead of in a hot loop.

    python tools/tune_in_step.py [--workload single] [--reps 3] [--min-gain 0.03] [--out gpurun_out/tile_table_instep.json]

tools/tune_tiles.py times every (tile, split-K) plan of a shape back to back on one operand set: L2 / Infinity-Cache-hot operands, no
neighbours.  In the step the operands were just produced by other kernels, and the ranking differs (a hot-loop retune of the pointwise
shapes made the training step 10 % slower, profiles/README.md round 5).  Here every shape of one eager UNet step (the launches a DDIM step
replays, HIP events around each, bench.eager_unet_step) runs trial plan j of its candidate list in pass j, all shapes at once; passes of
the committed table are interleaved as the reference.  A plan replaces the table's when its best-of-reps time beats the table's
best-of-reps time by --min-gain (and by >= 1 us per launch).  The result is a candidate table: A/B it with
LEFTREFILL_TILE_TABLE_PATH=<out> python bench.py before committing anything (split-K factors change fp32 rounding: run the GPU tests)."""
import argparse
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from leftrefill_amd import ops  # noqa: E402


def canon(plan):
    """(tile_m, tile_n, splits, pipe) with the pipe field as the table writes it: lr_gemm_plan reports the resolved LDS stage count (2 / 3)
    where the table says 0 = the tile's default instance; 4 = the deep 128-row ring, 8 = the halo-tile conv."""
    tm, tn, sp, stg = (tuple(plan) + (0,))[:4]
    return (int(tm), int(tn), int(sp), int(stg) if stg in (4, 8) else 0)


def greedy(a, unet, x, t, ctx, keys, base, per_cand):
    """Coordinate descent on the hipGraph-replayed step: shapes in order of their time per step, for each the plans that looked best inside
    the eager step; a plan is kept when the replayed step gets faster by more than --accept-ms, twice."""
    table = dict(ops.tile_cache())
    unet.use_hip_graph = True

    def step_ms(trial):
        ops.PLAN_TRIAL = dict(trial) if trial else None
        unet._graphs.clear()
        with torch.no_grad():
            unet(x, t, ctx)
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    unet(x, t, ctx)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4)
        ops.PLAN_TRIAL = None
        return best

    ref = [step_ms(None) for _ in range(3)]
    print("replayed step with the committed table:", " ".join(f"{v:.3f}" for v in ref), "ms", flush=True)
    best_ms = min(ref)
    accepted = {}
    order = sorted(keys, key=lambda k: -base[k][0])[:a.greedy_keys]
    for k in order:
        cur = canon(tuple(table[k])) if k in table else None
        if cur is None:
            continue
        cands = sorted((us, p) for p, us in per_cand[k].items() if p != cur)[:a.greedy_cands]
        for us, p in cands:
            trial = dict(accepted)
            trial[k] = p
            # (the box drifts by 0.1-0.3 ms over a run: every trial is compared with the accepted set measured right before it)
            b1 = step_ms(accepted)
            ms = step_ms(trial)
            if ms < b1 - a.accept_ms:
                b2 = step_ms(accepted)
                ms2 = step_ms(trial)
                if ms2 < b2 - a.accept_ms:
                    print(f"{k}: {list(cur)} -> {list(p)}   step {b1:.3f} / {b2:.3f} -> {ms:.3f} / {ms2:.3f} ms", flush=True)
                    accepted[k] = p
                    cur = p
    for k, p in accepted.items():
        table[k] = list(p)
    ref2 = [step_ms(None) for _ in range(2)]
    fin = [step_ms(accepted) for _ in range(2)]
    print("committed again:", " ".join(f"{v:.3f}" for v in ref2), "| accepted set:", " ".join(f"{v:.3f}" for v in fin), "ms")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({k: list(v) for k, v in sorted(table.items())}, f, indent=0)
    print("wrote", a.out, f"({len(accepted)} changed entries)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--greedy", action="store_true", help="second phase: coordinate descent on the replayed step (see greedy())")
    ap.add_argument("--greedy-keys", type=int, default=30)
    ap.add_argument("--greedy-cands", type=int, default=3)
    ap.add_argument("--accept-ms", type=float, default=0.015)
    ap.add_argument("--workload", default="single", choices=["single", "mv5", "train"],
                    help="train: one eager training step of configs[4] (bf16, batch 16 at 256x512: forward + input-gradient backward); first phase only")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--add-new", action="store_true", help="tabulate the shapes the table does not know (e.g. --batch 8)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--splits", type=int, nargs="*", default=[], help="explicit split-K factors to try besides 1 and the heuristic's (e.g. 2 3 4 6 8)")
    ap.add_argument("--min-gain", type=float, default=0.03)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_table_instep.json"))
    a = ap.parse_args()
    device = torch.device("cuda:0")
    train_step = None
    if a.workload == "train":
        # the step of bench.train_bench (NVS task model, prompt tokens trainable, bf16) without the optimizer: the GEMM launches of the
        # forward and of the input-gradient backward, in the order and cache state the training step runs them
        model = bench.build_model(device, "nvs").train()
        unet = model.model.diffusion_model
        unet.compute_dtype = torch.bfloat16
        for p_ in model.parameters():
            p_.requires_grad_(False)
        g = torch.Generator(device=device).manual_seed(1099)
        tokens = torch.nn.Parameter(0.02 * torch.randn(73, 1024, device=device, generator=g))
        base_ctx = torch.randn(16, 77, 1024, device=device, generator=g)
        c_cat = torch.randn(16, 5, 32, 64, device=device, generator=g)
        x_start = torch.randn(16, 4, 32, 64, device=device, generator=g)
        noise = torch.randn(16, 4, 32, 64, device=device, generator=g)
        t_buf = torch.randint(0, 1000, (16,), device=device, generator=g)

        def train_step(install):
            install()
            ctx_ = torch.cat([base_ctx[:, :1], base_ctx[:, 1:74] + tokens, base_ctx[:, 74:]], dim=1)
            loss, _ = model.p_losses(x_start, {"c_concat": [c_cat], "c_crossattn": [ctx_]}, t_buf, noise=noise)
            loss.backward()
            tokens.grad = None
        x = t = ctx = None
    else:
        model = bench.build_model(device, a.workload)
        unet = model.model.diffusion_model
        B = a.batch
        c_concat, c_cross, uc_cross, x_T = bench.synthetic_batch(B, 64, 128, device, 7)
        x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1)
        t = torch.full((2 * B,), 501, device=device, dtype=torch.long)
        ctx = torch.cat([uc_cross, c_cross]).half()
    unet.use_hip_graph = False

    events = []          # (key, e0, e1, plan)
    state = {"key": None, "e0": None, "plan": None}

    def hook(key, phase, info):
        if phase == 0:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            state.update(key=key, e0=e0, plan=info)
        else:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            if info == 0:
                events.append((key, state["e0"], e1, state["plan"]))

    def run_pass():
        """one eager step -> {key: (total us, launches, resolved plan)}"""
        events.clear()
        ops.LAUNCH_HOOK = None
        if train_step is not None:
            train_step(lambda: setattr(ops, "LAUNCH_HOOK", hook))
        else:
            with torch.no_grad():
                bench.eager_unet_step(unet, x, t, ctx, hook=lambda: setattr(ops, "LAUNCH_HOOK", hook))
        torch.cuda.synchronize()
        ops.LAUNCH_HOOK = None
        res = defaultdict(lambda: [0.0, 0, None])
        for key, e0, e1, plan in events:
            r = res[key]
            r[0] += 1e3 * e0.elapsed_time(e1)
            r[1] += 1
            r[2] = plan
        return res

    ops.PLAN_TRIAL = None
    run_pass()                                   # warm-up (packs, caches)
    base = run_pass()
    keys = sorted(base)
    print(f"{len(keys)} GEMM shapes, {sum(v[1] for v in base.values())} launches per step", flush=True)
    cands = []
    for tm, tn, stg in ops.TILE_CANDIDATES:
        for sp in (1, 0) + tuple(a.splits):
            cands.append((tm, tn, sp, stg))
    best_ref = {k: float("inf") for k in keys}                       # table plan: best of all reference passes
    best_trial = {k: (float("inf"), None) for k in keys}             # (us, resolved plan)
    per_cand = {k: {} for k in keys}                                 # canonical plan -> best us of the shape's launches in one step

    def fold_ref(res):
        for k in keys:
            if k in res:
                best_ref[k] = min(best_ref[k], res[k][0])

    fold_ref(base)
    for j, cand in enumerate(cands):
        for rep in range(a.reps):
            ops.PLAN_TRIAL = {k: cand for k in keys}
            try:
                res = run_pass()
            except Exception as e:      # a helper refused the plan for the shape in flight: drop the trial for it and go on
                bad = state["key"]
                print(f"  trial {cand}: {type(e).__name__} at {bad}", flush=True)
                torch.cuda.synchronize()
                ops.LAUNCH_HOOK = None
                break
            trial_map = ops.PLAN_TRIAL
            for k in keys:
                if k in res and trial_map.get(k) is not None and res[k][2] is not None:
                    plan = tuple(int(v) for v in res[k][2])
                    if res[k][0] < best_trial[k][0]:
                        best_trial[k] = (res[k][0], plan)
                    pc = per_cand[k].setdefault(canon(plan), float("inf"))
                    per_cand[k][canon(plan)] = min(pc, res[k][0])
        ops.PLAN_TRIAL = None
        fold_ref(run_pass())
        print(f"trial {j + 1}/{len(cands)} {cand} done", flush=True)
    ops.PLAN_TRIAL = None
    if a.greedy and train_step is None:
        return greedy(a, unet, x, t, ctx, keys, base, per_cand)
    table = dict(ops.tile_cache())
    gain = 0.0
    for k in keys:
        ref, (tr, plan) = best_ref[k], best_trial[k]
        n = base[k][1]
        cur = tuple(table.get(k, ())) if k in table else None
        if plan is not None and cur is None and a.add_new:
            # a shape the table does not know (another batch size): tabulate the better of the refined heuristic's plan and the best trial
            hp = canon(base[k][2]) if base[k][2] is not None else None
            take = canon(plan) if (hp is None or (tr < ref * (1.0 - a.min_gain) and (ref - tr) / n >= 1.0)) else hp
            print(f"{k}: new, heuristic {list(hp) if hp else None} {ref / n:7.1f} us, best trial {list(canon(plan))} {tr / n:7.1f} us  x{n} -> {list(take)}", flush=True)
            table[k] = list(take)
            if take != hp:
                gain += ref - tr
            continue
        if plan is None or cur is None:
            continue
        cur4 = tuple(cur) + ((0,) if len(cur) == 3 else ())
        if canon(plan) != canon(cur4) and tr < ref * (1.0 - a.min_gain) and (ref - tr) / n >= 1.0:
            print(f"{k}: {list(cur4)} {ref / n:7.1f} us -> {list(plan)} {tr / n:7.1f} us  x{n}  ({ref - tr:+.1f} us per step)", flush=True)
            table[k] = list(canon(plan))
            gain += ref - tr
    print(f"proposed changes save {gain:.1f} us per eager step (sum of per-shape best-of-{a.reps})")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({k: list(v) for k, v in sorted(table.items())}, f, indent=0)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
