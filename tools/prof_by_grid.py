#!/usr/bin/env python3
"""Per (kernel, grid) breakdown of a rocprofv3 rocpd database: tells the GEMM shapes of a step apart.
usage: prof_by_grid.py results.db [steps] [name-substring]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# a training profile counts its own steps: one adamw_kernel launch per optimizer step (the warm-up / settle steps a
# command runs besides its --steps are in the trace too; round 3 divided a C4 trace of 8 steps by the 7 on its command line)
_n = con.execute("select count(*) from kernels where name like '%adamw_kernel%'").fetchone()[0]
if _n > 0:
    steps = _n
pat = sys.argv[3] if len(sys.argv) > 3 else "gemm_kernel"
rows = con.execute("select name, grid_x, grid_y, count(*), sum(end-start), avg(end-start) from kernels where name like ? "
                   "group by name, grid_x, grid_y order by 5 desc", (f"%{pat}%",)).fetchall()
tot = con.execute("select sum(end-start) from kernels").fetchone()[0]
print(f"{'% step':>7} {'calls/step':>10} {'avg us':>8}  grid (workgroups)      kernel")
for r in rows[:40]:
    n = re.sub(r"smx::", "", r[0]); n = re.sub(r"\(.*\)$", "", n); n = re.sub(r"^void ", "", n)
    print(f"{r[4]/tot*100:7.2f} {r[3]/steps:10.1f} {r[5]/1e3:8.1f}  {r[1]//256:>8d} x {r[2]:<6d}  {n[:70]}")
