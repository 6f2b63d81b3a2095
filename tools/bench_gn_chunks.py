"""What does the in-kernel statistics prologue of gn_apply cost as the number of partial chunks grows?  (Decides whether producers
that emit per-GROUP sums per row block could drop the gn_finalize launch: DESIGN section 5.)

    python tools/bench_gn_chunks.py [N HW C]

Times lr_groupnorm_apply_n inside a hipGraph (rotating > 256 MB of operands) with partials [N][nchunks][32][2] for several nchunks,
and gn_finalize + apply(nchunks = 1) as used today."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leftrefill_amd import _lib, ops  # noqa: E402

N, HW, C = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (8, 8192, 320)
dev = torch.device("cuda:0")
lib = _lib.load()
nsets = max(2, int(600e6 / (N * HW * C * 2 * 2)) + 1)
xs = [torch.randn(N * HW, C, device=dev).half() for _ in range(nsets)]
ys = [torch.empty_like(x) for x in xs]
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
p = lambda t: t.data_ptr()


def timed(fn, reps=3):
    fn(0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(2 * nsets):
            fn(i % nsets)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / (2 * nsets))
    return best


for nch in (1, 8, 16, 32, 64, 128, 256):
    part = torch.rand(N, nch, 32, 2, device=dev) * (HW * C / 32 / nch)
    part[..., 1] += part[..., 0] ** 2 / (HW * C / 32 / nch)

    def fn(i):
        st = ops._stream()
        _lib.check(lib.lr_groupnorm_apply_n(p(xs[i]), C, None, 0, N, HW, p(part), nch, p(g), p(b), 1e-5, 1, p(ys[i]), st), "apply_n")
    print(f"apply_n N={N} HW={HW} C={C} nchunks={nch:4d}: {timed(fn):6.2f} us")

for R in (64, 128, 256):
    gs = torch.rand(N * HW // R, C, 2, device=dev)
    part = torch.empty(N, 1, 32, 2, device=dev)

    def fn2(i):
        st = ops._stream()
        _lib.check(lib.lr_groupnorm_finalize(p(gs), C, R, None, 0, 1, N, HW, p(part), st), "finalize")
        _lib.check(lib.lr_groupnorm_apply_n(p(xs[i]), C, None, 0, N, HW, p(part), 1, p(g), p(b), 1e-5, 1, p(ys[i]), st), "apply_n")
    print(f"finalize(R={R}) + apply_n(1): {timed(fn2):6.2f} us")
