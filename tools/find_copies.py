#!/usr/bin/env python3
"""Which Python lines of a training step trigger aten copy / clone / fill kernels (they are launch overhead at small batch)?"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from summarymixing_amd.trainer import FlatAdamW
cfg = dict(bench.CONFIGS["c2b"]); cfg["B"] = 16
enc = bench.build_encoder(cfg, torch.device("cuda"), 0.15)
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, torch.device("cuda"), torch.bfloat16)
def step():
    opt.zero_grad(); enc(src, wav_len).backward(r); opt.step()
for _ in range(3): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::ne", "aten::mul", "aten::add", "aten::round", "aten::lt", "aten::arange"):
        st = [f for f in (ev.stack or []) if "summarymixing_amd" in f or "bench.py" in f]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d}  {name:18s} {where}")
