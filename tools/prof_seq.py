#!/usr/bin/env python3
"""Kernel launch sequence of ONE step out of a rocprofv3 rocpd database (kernel-trace): run-length encoded names, in start order.
usage: prof_seq.py results.db [pattern-to-highlight]   (step boundaries: adamw_kernel)"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else None
rows = con.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
a, b = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
seq = rows[a:b]
short = lambda n: re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", re.sub(r"smx::", "", n)))[:70]
print(f"# {len(seq)} launches in the last complete step; span {(seq[-1][2] - seq[0][1]) / 1e3:.1f} us, kernel time {sum(r[2] - r[1] for r in seq) / 1e3:.1f} us")
out, prev, cnt = [], None, 0
for i, r in enumerate(seq):
    n = short(r[0])
    if pat and pat in n:
        lo = max(0, i - 2)
        print("  ..", " | ".join(short(x[0]) for x in seq[lo:i + 2]))
