#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) per kernel: calls, total, average.
usage: prof_summary.py results.db [steps]   -> markdown-ish table on stdout (commit it under profiles/)"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# a training profile counts its own steps: one adamw_kernel launch per optimizer step (the warm-up / settle steps a
# command runs besides its --steps are in the trace too; round 3 divided a C4 trace of 8 steps by the 7 on its command line)
_n = con.execute("select count(*) from kernels where name like '%adamw_kernel%'").fetchone()[0]
if _n > 0:
    steps = _n
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step")
print(f"{'%':>6} {'calls/step':>10} {'avg us':>9} {'min us':>8} {'max us':>8}  kernel")
for r in rows[:28]:
    n = re.sub(r"smx::", "", r[0]); n = re.sub(r"\(.*\)$", "", n); n = re.sub(r"^void ", "", n)
    print(f"{r[2]/tot*100:6.1f} {r[1]/steps:10.1f} {r[3]/1e3:9.1f} {r[4]/1e3:8.1f} {r[5]/1e3:8.1f}  {n[:100]}")
# launches BEFORE the first kernel of the first step belong to the model set-up (parameters copied into the flat fp32 / bf16 buffers
# by FlatAdamW: one __amd_rocclr_copyBuffer per parameter): say so, because calls/step above divides them by the step count
try:
    t0 = con.execute("select min(start) from kernels where name like '%gemm_kernel%' or name like '%layernorm%'").fetchone()[0]
    pre = con.execute("select count(*), sum(end-start) from kernels where start < ?", (t0,)).fetchone()
    cp = con.execute("select count(*) from kernels where name like '%copyBuffer%'").fetchone()[0]
    cp_pre = con.execute("select count(*) from kernels where name like '%copyBuffer%' and start < ?", (t0,)).fetchone()[0]
    print(f"# set-up (before the first step's first kernel): {pre[0]} launches, {(pre[1] or 0)/1e6:.2f} ms; __amd_rocclr_copyBuffer: {cp_pre} of {cp} launches are set-up "
          f"(= {(cp - cp_pre) / steps:.1f} per step)")
except Exception:
    pass
