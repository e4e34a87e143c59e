#!/usr/bin/env python3
"""Effective shader clock per kernel = GRBM_GUI_ACTIVE / kernel wall time, from a `rocprofv3 --pmc GRBM_GUI_ACTIVE
--kernel-trace --output-format csv` run.  usage: grbm_clock.py OUTDIR [top]"""
import collections
import csv
import glob
import sys

d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
cc = [r for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(fn))]
kt = {}
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        kt[r.get("Dispatch_Id")] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in cc:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    elif r.get("Dispatch_Id") in kt:
        t0, t1 = kt[r["Dispatch_Id"]]
    else:
        continue
    name = r["Kernel_Name"].split("(")[0][:70] + " grid " + r.get("Grid_Size", "?")
    a = acc[name]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (t1 - t0)
tot = sum(a[2] for a in acc.values()) or 1
print(f"# {len(cc)} counter rows; GRBM_GUI_ACTIVE / wall ns = GHz (if the counter sums the 8 XCDs the figure is 8x the clock: see the /8 column)")
print(f"{'share':>6s} {'calls':>6s} {'avg us':>8s} {'cyc/ns':>7s} {'/8':>6s}  kernel")
for name, a in sorted(acc.items(), key=lambda kv: -kv[1][2])[:top]:
    print(f"{100 * a[2] / tot:6.1f} {a[0]:6d} {a[2] / a[0] / 1e3:8.1f} {a[1] / a[2]:7.3f} {a[1] / a[2] / 8:6.3f}  {name}")
