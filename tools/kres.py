#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.  usage: kres.py remarks.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("spill", r"VGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("LDS", r"LDS Size \[bytes/block\]"), ("occ", r"Occupancy \[waves/SIMD\]"), ("SGPR", r"SGPRs")]
for b in txt.split("Function Name: ")[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("smx::", "").replace("void ", "").replace("__hip_bfloat16", "bf16").replace("(anonymous namespace)::", "")
    if flt and flt not in dem:
        continue
    vals = []
    for k, pat in keys:
        m = re.search(pat + r": (\d+)", b)
        vals.append(f"{k} {m.group(1) if m else '?':>5}")
    print(f"{dem[:80]:80s} " + "  ".join(vals))
