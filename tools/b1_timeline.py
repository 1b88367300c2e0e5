"""Kernel timeline of the last forward in a `rocprofv3 --kernel-trace` database of tools/b1_forward.py:
    python tools/b1_timeline.py gpurun_out/prof_b1/b1_results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ks = list(db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
first = [i for i, k in enumerate(ks) if "conv_first" in k[0] or "<1, 4, 0, 1>" in k[0] or ("conv16_f16" in k[0] and "true, true" in k[0])]
if len(first) < 2:
    sys.exit("fewer than two forwards in the trace")
seq = ks[first[-2]:first[-1]]
t0, tot = seq[0][1], 0
for n, st, en, g, w in seq:
    short = re.sub(r"adn::\(anonymous namespace\)::", "", n)
    short = re.sub(r"\(.*", "", short)[:70]
    print(f"{(st - t0) / 1e3:8.1f} us  dur {(en - st) / 1e3:7.1f}  grid {g // w:5d} x {w:4d}  {short}")
    tot += en - st
print(f"sum of kernels {tot / 1e3:.1f} us; span {(ks[first[-1]][1] - t0) / 1e3:.1f} us")
