#!/usr/bin/env python3
"""Run smx_gemm launches in a loop for a fixed wall time (the workload side of tools/power_trace.sh).

usage: loop_gemm.py SECONDS "LAYOUT N K M [epi]" ["LAYOUT N K M [epi]" ...]
Several shapes = a chain: they are launched round robin, back to back, as the layers of a step would.  Prints the average
duration per launch of each shape (HIP events over the whole run)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import build  # noqa: E402

secs = float(sys.argv[1])
specs = [s.split() for s in sys.argv[2:]]
fns = []
for sp in specs:
    fn, nb = build(int(sp[1]), int(sp[2]), int(sp[3]), sp[0], epi=sp[4] if len(sp) > 4 else "swishz")
    fns.append(fn)
for fn in fns:
    fn()
torch.cuda.synchronize()
if len(fns) == 1:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(200):
            fns[0]()
        n += 200
        torch.cuda.synchronize()
    e1.record()
    e1.synchronize()
    print(f"{' '.join(specs[0]):40s} {e0.elapsed_time(e1) * 1e3 / n:8.1f} us per launch over {n} launches", flush=True)
else:
    # chain: per-shape time from events around each launch of every 50th round (the others run unbracketed)
    acc = [0.0] * len(fns)
    cnt, rounds, t0 = 0, 0, time.time()
    while time.time() - t0 < secs:
        for r in range(50):
            if r == 49:
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(fns) + 1)]
                evs[0].record()
                for i, fn in enumerate(fns):
                    fn()
                    evs[i + 1].record()
                evs[-1].synchronize()
                for i in range(len(fns)):
                    acc[i] += evs[i].elapsed_time(evs[i + 1])
                cnt += 1
            else:
                for fn in fns:
                    fn()
        rounds += 50
        torch.cuda.synchronize()
    for sp, a in zip(specs, acc):
        print(f"{' '.join(sp):40s} {a * 1e3 / cnt:8.1f} us per launch (in the chain, {rounds} rounds)", flush=True)
