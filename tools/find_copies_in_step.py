#!/usr/bin/env python3
"""Which host-side calls of one C2b training step end in a device memcpy (torch profiler, grouped by Python stack)?"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from summarymixing_amd.trainer import FlatAdamW  # noqa: E402

cfg = dict(bench.CONFIGS["c2b"])
cfg["B"] = int(os.environ.get("B", "32"))
dev = torch.device("cuda", 0)
enc = bench.build_encoder(cfg, dev, 0.15)
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, dev, torch.bfloat16)


def step():
    opt.zero_grad()
    enc(src, wav_len).backward(r)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
stacks = collections.Counter()
for ev in prof.events():
    cnt[ev.name] += 1
    if (ev.name.startswith("hipMemcpy") or ev.name.startswith("hipMemset") or "Memcpy" in ev.name or ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_")) and ev.stack:
        own = [fr for fr in ev.stack if "/repo/" in fr and "find_copies" not in fr][:3]
        stacks[(ev.name, " <- ".join(own))] += 1
print("# host-side copy / fill calls of one step by the repo frames that issued them")
for (n, st), c in stacks.most_common(40):
    print(f"{c:5d}  {n:18s} {st}")
print("# all event names")
for n, c in cnt.most_common(60):
    if not n.startswith("void smx") and "smx::" not in n:
        print(f"{c:5d}  {n[:110]}")
