#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel-trace database: over the LAST `steps` optimizer steps (delimited by
adamw_kernel launches) the union of the busy intervals of all streams against the wall time, and the histogram of the gaps.
usage: prof_gaps.py results.db [steps]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ad = [r[0] for r in con.execute("select end from kernels where name like '%adamw_kernel%' order by start").fetchall()]
if len(ad) < steps + 1:
    steps = len(ad) - 1
t0, t1 = ad[-steps - 1], ad[-1]
rows = con.execute("select start, end from kernels where start >= ? and end <= ? order by start", (t0, t1)).fetchall()
busy, gaps, cur_s, cur_e = 0, [], None, None
for s, e in rows:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
wall = t1 - t0
print(f"{steps} steps: wall {wall/1e6/steps:.3f} ms/step, some kernel running {busy/1e6/steps:.3f} ms/step, idle {(wall-busy)/1e6/steps:.3f} ms/step "
      f"({100*(wall-busy)/wall:.1f} %), {len(rows)/steps:.0f} launches and {len(gaps)/steps:.0f} gaps per step")
sumk = sum(e - s for s, e in rows)
print(f"sum of kernel durations {sumk/1e6/steps:.3f} ms/step (overlap of streams {(sumk-busy)/1e6/steps:.3f} ms/step)")
import collections
h = collections.Counter()
for g in gaps:
    b = 0.5 if g < 500 else 1 if g < 1000 else 2 if g < 2000 else 4 if g < 4000 else 8 if g < 8000 else 16 if g < 16000 else 99
    h[b] += g
for b in sorted(h):
    n = sum(1 for g in gaps if (0.5 if g < 500 else 1 if g < 1000 else 2 if g < 2000 else 4 if g < 4000 else 8 if g < 8000 else 16 if g < 16000 else 99) == b)
    print(f"  gaps < {b if b != 99 else 'inf'} us: {n/steps:7.1f} per step, {h[b]/1e6/steps:.3f} ms/step")
