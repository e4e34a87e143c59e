import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity
cfg = dict(bench.CONFIGS["c2b"]); cfg["layers"] = 2
enc = bench.build_encoder(cfg, "cuda", 0.15)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, "cuda", torch.bfloat16)
from summarymixing_amd.trainer import FlatAdamW
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
def step():
    opt.zero_grad(); y = enc(src, wav_len); y.backward(r); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if "emcpy" in e.name or "copy_" in e.name or "aten::to" == e.name or "aten::contiguous" in e.name or "aten::clone" in e.name]
from collections import Counter
c = Counter()
for e in evs:
    st = [f for f in (e.stack or []) if "summarymixing_amd" in f or "bench" in f or "find_copies" in f]
    c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(40): print(v, k)
