#!/usr/bin/env python3
"""Summarise a tools/power_trace CSV: clock / power statistics of the loaded window.
usage: power_summary.py trace.csv [label]   (loaded = gfx_busy >= 50 % or power >= 60 % of the trace's maximum)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
if not rows:
    print(f"{label}: empty trace")
    sys.exit(0)
f = lambda r, k: float(r[k])
pmax = max(f(r, "power_w") for r in rows)
busy = [r for r in rows if f(r, "power_w") >= 0.6 * pmax]
if len(busy) > 20:
    busy = busy[len(busy) // 10:]            # drop the ramp
def st(rs, k):
    v = sorted(f(r, k) for r in rs)
    n = len(v)
    return v[0], v[n // 10], v[n // 2], v[9 * n // 10], v[-1], sum(v) / n
dt = (f(rows[-1], "t_ms") - f(rows[0], "t_ms")) / max(len(rows) - 1, 1)
print(f"## {label}: {len(rows)} firmware samples, one per {dt:.2f} ms; loaded window {len(busy)} samples "
      f"({f(busy[0], 't_ms'):.0f} .. {f(busy[-1], 't_ms'):.0f} ms)")
for k, unit in (("gfxclk_mhz", "MHz"), ("gfxclk_min", "MHz"), ("gfxclk_max", "MHz"), ("uclk_mhz", "MHz"), ("power_w", "W"), ("gfx_busy", "%"), ("umc_busy", "%"), ("temp_hot", "C")):
    mn, p10, med, p90, mx, mean = st(busy, k)
    print(f"  {k:11s} min {mn:7.0f}  p10 {p10:7.0f}  median {med:7.0f}  p90 {p90:7.0f}  max {mx:7.0f}  mean {mean:8.1f} {unit}")
idle = [r for r in rows if f(r, "power_w") < 0.3 * pmax]
if idle:
    print(f"  idle samples: {len(idle)}; gfxclk median {st(idle, 'gfxclk_mhz')[2]:.0f} MHz, power median {st(idle, 'power_w')[2]:.0f} W")
for k in ("ppt_acc", "thm_acc"):
    a, b = int(busy[0][k]), int(busy[-1][k])
    acc0, acc1 = int(rows[0][k]), int(rows[-1][k])
    print(f"  {k}: {b - a} over the loaded window ({acc1 - acc0} over the trace)")
e0, e1 = int(busy[0]["energy"]), int(busy[-1]["energy"])
tw = (f(busy[-1], "t_ms") - f(busy[0], "t_ms")) * 1e-3
if tw > 0:
    print(f"  energy accumulator: {(e1 - e0) * 15.3e-6 / tw:.0f} W average over the loaded window (15.3 uJ units)")
