#!/usr/bin/env python3
"""Every kernel launch of the LAST complete step of a rocprofv3 rocpd database, in start order: index, start offset (us), duration (us),
gap to the previous launch's end (us), grid (workgroups), short name.   usage: prof_fullseq.py results.db   (step boundary: adamw_kernel)"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gx}, {wx}" if gx and wx else "")
rows = con.execute(f"select {sel} from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
a, b = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
seq = rows[a:b]
short = lambda n: re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", re.sub(r"smx::", "", n)))[:90]
t0 = seq[0][1]
print(f"# {len(seq)} launches; span {(seq[-1][2] - t0) / 1e3:.1f} us; kernel time {sum(r[2] - r[1] for r in seq) / 1e3:.1f} us")
prev_end = t0
for i, r in enumerate(seq):
    wg = (r[3] // r[4]) if len(r) > 3 and r[4] else 0
    print(f"{i:4d} {(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} {(r[1] - prev_end) / 1e3:6.1f} {wg:6d}  {short(r[0])}")
    prev_end = max(prev_end, r[2])
