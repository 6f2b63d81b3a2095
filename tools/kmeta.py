"""Print per-kernel resource usage (VGPRs, spills, SGPRs, scratch, LDS) from a hipcc --save-temps .s file."""
import re
import sys

s = open(sys.argv[1]).read()
md = s[s.index("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(re.escape(k) + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    print("%-72s vgpr %4s agpr %4s spill %3s sgpr %4s scratch %4s lds %s" % (
        g(".name")[:72], g(".vgpr_count"), blk.split()[0], g(".vgpr_spill_count"), g(".sgpr_count"),
        g(".private_segment_fixed_size"), g(".group_segment_fixed_size")))
