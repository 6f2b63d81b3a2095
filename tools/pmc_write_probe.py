"""Where do a kernel's output bytes leave the L2s?  python tools/pmc_write_probe.py <dir of a `rocprofv3 --pmc WRITE_SIZE` pass over tools/pmc_step.py run>

Per-dispatch WRITE_SIZE (KiB -> MB) of the conv_halo<320> dispatches and of the dispatch right after each, grouped by that follower (level-0
convs with the full epilogue write 65536 x 320 x 2 B = 41.9 MB; K-slice launches write fp32 partials).  Compares builds with plain and
write-through epilogue stores.  Under --pmc every dispatch runs alone with idle time around it: the numbers say in whose counter window the
bytes left the L2s, not what happens between back-to-back kernels of a graph replay."""
import statistics
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import pmc_step

rows = [r for r in pmc_step._rows(sys.argv[1]) if r["Counter_Name"] == "WRITE_SIZE"]
mb = lambda r: float(r["Counter_Value"]) * 1024.0 / 1e6
# conv_halo<320> dispatches of the whole run (two eager steps), split by what follows them: gn_apply = a level-0 conv with the full epilogue
# (41.9 MB of output), splitk_reduce = a K-slice launch that wrote fp32 partials (plain stores in every build)
groups = defaultdict(lambda: ([], []))
for a, b in zip(rows, rows[1:]):
    if "conv_halo_kernelILi320" in a["Kernel_Name"]:
        g = groups["gn_apply" if "gn_apply" in b["Kernel_Name"] else "splitk_reduce" if "splitk_reduce" in b["Kernel_Name"] else b["Kernel_Name"][:30]]
        g[0].append(mb(a))
        g[1].append(mb(b))
for k, (own, nxt) in groups.items():
    print(f"conv_halo<320> followed by {k:14s} n={len(own):2d}  own WRITE_SIZE mean {statistics.mean(own):6.1f}  min {min(own):6.1f}  max {max(own):6.1f} MB"
          f" | follower's WRITE_SIZE mean {statistics.mean(nxt):6.1f}  min {min(nxt):6.1f}  max {max(nxt):6.1f} MB")
