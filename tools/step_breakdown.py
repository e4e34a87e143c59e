#!/usr/bin/env python3
"""Every libsmx launch of one instrumented training step, grouped by name (HIP events on the launch streams):
    python tools/step_breakdown.py [c2b|c2a|c4] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from summarymixing_amd import ops
from summarymixing_amd.trainer import FlatAdamW
name = sys.argv[1] if len(sys.argv) > 1 else "c2b"
cfg = dict(bench.CONFIGS[name])
if len(sys.argv) > 2:
    cfg["B"] = int(sys.argv[2])
dev = torch.device("cuda")
enc = bench.build_encoder(cfg, dev, 0.15)
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, dev, torch.bfloat16)
def step():
    opt.zero_grad(); enc(src, wav_len).backward(r); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
ops.prof_start(); step(); recs = ops.prof_stop()
tot = sum(x[3] for x in recs)
agg = {}
for nm, nb, fl, ms in recs:
    e = agg.setdefault(nm, [0, 0.0, 0.0]); e[0] += 1; e[1] += ms; e[2] += nb
print(f"{name}: {len(recs)} launches, {tot:.2f} ms of kernel time")
for nm, (c, ms, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{ms/tot*100:5.1f}%  {c:4d} x {ms*1e3/c:7.1f} us  {nb/ms/1e6 if ms else 0:6.0f} GB/s  {nm}")
