"""Experiment (MI355X): ONE UNet step of configs[1] (batch 8) as a single captured stream vs the two batch halves on two streams inside one
hipGraph (every GEMM planned like the full batch: same tiles, bit-identical rows), optionally with the second stream starting late.

    python tools/exp_stream_split.py [--delay-us 0,40,80]

Idea under test: all 256 blocks of a launch run their load / multiply / store phases in lockstep, so the chip alternates between
HBM-bound and MFMA-bound phases; two half-chip launches that are out of phase could overlap them."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from leftrefill_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--delay-us", default="0,30,60,120")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, "single")
    unet = model.model.diffusion_model
    B, h, w = 4, 64, 128
    c_concat, c_cross, uc_cross, x_T = bench.synthetic_batch(B, h, w, dev, 7)
    x = torch.cat([torch.cat([x_T] * 2), torch.cat([c_concat] * 2)], dim=1).float().contiguous()
    t = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
    ctx = torch.cat([uc_cross, c_cross]).half().contiguous()
    N = 2 * B
    hN = N // 2
    with torch.no_grad():
        unet.prepare()
        kv = unet._context_kv(ctx)
        emb = unet._embed(t).contiguous()
        L = ctx.shape[1]

        def kv_half(i):
            sl = slice(i * hN * L, (i + 1) * hN * L)
            out = []
            for ent in kv:
                e = [ent[0][sl], None if ent[1] is None else ent[1][i * hN:(i + 1) * hN]]
                if len(ent) > 2:
                    e += [ent[2][sl], ent[3][i * hN:(i + 1) * hN]]
                out.append(tuple(e))
            return out

        def full():
            return unet._run_plan(x, t, ctx, kv, False, emb_all=emb)

        kvh = [kv_half(0), kv_half(1)]
        s2 = torch.cuda.Stream()
        spin = torch.empty(1 << 20, device=dev)

        def split(delay_us):
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with ops.plan_batch_scale(2):
                with torch.cuda.stream(s2):
                    if delay_us:
                        torch.cuda._sleep(int(delay_us * 2100))      # ~cycles at 2.1 GHz
                    o1 = unet._run_plan(x[hN:], t[hN:], ctx[hN:], kvh[1], False, emb_all=emb[hN:])
                o0 = unet._run_plan(x[:hN], t[:hN], ctx[:hN], kvh[0], False, emb_all=emb[:hN])
            cur.wait_stream(s2)
            return torch.cat([o0, o1])

        def capture(fn):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            return g, out

        def timeit(g, n=20):
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(n):
                    g.replay()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n * 1e3)
            return best

        g_full, o_full = capture(full)
        ms_full = timeit(g_full)
        print(f"one stream, batch {N}: {ms_full:.3f} ms per UNet step", flush=True)
        for d in [int(v) for v in a.delay_us.split(",")]:
            g_s, o_s = capture(lambda: split(d))
            ms = timeit(g_s)
            g_s.replay()
            torch.cuda.synchronize()
            g_full.replay()
            torch.cuda.synchronize()
            same = torch.equal(o_s, o_full)
            rel = ((o_s.float() - o_full.float()).norm() / o_full.float().norm()).item()
            print(f"two streams (halves of {hN}), second stream delayed {d:4d} us: {ms:.3f} ms  ({100 * (ms / ms_full - 1):+.1f} %)  bit-identical {same} rel {rel:.2e}",
                  flush=True)
        ms_full2 = timeit(g_full)
        print(f"one stream again: {ms_full2:.3f} ms")


if __name__ == "__main__":
    main()
