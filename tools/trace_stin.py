"""Developer tool: per-piece timeline of stin_block_kernel from in-kernel shader-clock stamps (trace build of the library:
VARIANT_SRCS="stin_block elementwise" tools/build_variant.sh sitrace -DSI_TRACE; LEFTREFILL_LIB_PATH=.../libleftrefill_hip_sitrace.so).

Stamps per (block, wave): 0 start; per piece i (0 .. 19): 2 + 3 i = reached the piece's wait, 3 + 3 i = vmcnt wait passed, 4 + 3 i = barrier
passed; 62 / 63 = before / after the vmcnt(0) in front of the x1 reload; 1 = all pieces done."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from leftrefill_amd import _lib, ops, packing  # noqa: E402


def main():
    lib = _lib.load()
    lib.lr_stin_set_trace.argtypes = [ctypes.c_void_p]
    lib.lr_stin_set_trace.restype = None
    d = torch.device("cuda:0")
    C, NQ, M = 320, 960, 65536
    wp = (torch.randn(C, C) / C ** 0.5).half().to(d)
    bp = torch.randn(C).to(d)
    wf, bf, _ = [t.to(d) for t in packing.fold_layernorm(torch.randn(NQ, C) / C ** 0.5, None, torch.ones(C), torch.zeros(C))]
    xs = [torch.randn(M, C, device=d).half() for _ in range(8)]
    run = lambda x: ops.stin_block(x, wp, bp, wf, bf, eps=1e-5)
    for x in xs[:4]:
        run(x)
    torch.cuda.synchronize()
    nb = M // 256
    trace = torch.zeros(nb * 8 * 64, device=d, dtype=torch.int64)
    lib.lr_stin_set_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(xs[5])
    e1.record()
    torch.cuda.synchronize()
    lib.lr_stin_set_trace(None)
    tr = trace.reshape(nb, 8, 64).cpu().double()
    t0 = tr[:, :, 0].min()
    print(f"kernel {1e3 * e0.elapsed_time(e1):.1f} us; span {(tr[:, :, 1].max() - t0).item():.0f} ticks of the shader clock counter; blocks {nb}")
    seg = lambda a, b_: (tr[:, :, b_] - tr[:, :, a]).mean().item()
    print(f"start skew {(tr[:, :, 0] - t0).mean().item():8.0f}   start -> first piece's wait (row loads issued, bias rows, ring prologue) {seg(0, 2):7.0f}")
    tot = {"vmcnt": 0.0, "barrier": 0.0, "body": 0.0}
    for i in range(20):
        w, b_, g = 2 + 3 * i, 3 + 3 * i, 4 + 3 * i
        nxt = 2 + 3 * (i + 1) if i < 19 else 1
        if i == 4:
            nxt = 62
        body = seg(g, nxt)
        print(f"  piece {i:2d}: vmcnt wait {seg(w, b_):6.0f}  barrier {seg(b_, g):6.0f}  k-loop (+ previous piece's epilogue) {body:6.0f}")
        tot["vmcnt"] += seg(w, b_); tot["barrier"] += seg(b_, g); tot["body"] += body
        if i == 4:
            print(f"  between the stages: last epilogue + vmcnt(0) {seg(62, 63):6.0f}   reload + LayerNorm {seg(63, 2 + 3 * 5):6.0f}")
    print(f"totals: vmcnt waits {tot['vmcnt']:.0f}  barriers {tot['barrier']:.0f}  bodies {tot['body']:.0f}  whole {seg(0, 1):.0f}")


if __name__ == "__main__":
    main()
