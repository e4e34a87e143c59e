"""Which parameter gradients differ between (a) accumulated backward passes, (b) the sum of separately computed gradients and
(c) the fused batch?  usage: accum_diag.py [fp32|bf16] [ragged]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from summarymixing_amd.trainer import FlatAdamW, fuse_microbatches
from summarymixing_amd import functional as F
from test_accum_gpu import _micro
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
ragged = len(sys.argv) > 2
cfg = dict(bench.CONFIGS["c1"])
enc = bench.build_encoder(cfg, torch.device("cuda"), 0.0)
opt = FlatAdamW(enc, lr=1e-3, compute_dtype=dtype)
micro = [_micro(cfg, 3, 60, 1, dtype), _micro(cfg, 2, 48 if ragged else 60, 2, dtype)]

def grads(batches):
    opt.zero_grad()
    for x, wl, r in batches:
        enc(x, wl).backward(r)
    F.flush_deferred(); F.join_side(); torch.cuda.synchronize()
    return opt.flat_g.clone()

g_acc = grads(micro)
g_sum = grads(micro[:1]) + grads(micro[1:])
xs, wls = fuse_microbatches([(m[0], m[1]) for m in micro])
T = xs.shape[1]
rs = torch.cat([torch.nn.functional.pad(m[2], (0, 0, 0, T - m[2].shape[1])) for m in micro])
g_fus = grads([(xs, wls, rs)])
names = dict((id(p), n) for n, p in enc.named_parameters())
print(f"{'parameter':70s} acc-vs-sum  fused-vs-sum")
for p in enc.parameters():
    a, b = opt.param_range([p])
    s = g_sum[a:b]; m = float(s.abs().max()) + 1e-30
    e1, e2 = float((g_acc[a:b] - s).abs().max()) / m, float((g_fus[a:b] - s).abs().max()) / m
    flag = " <--" if max(e1, e2) > 1e-3 else ""
    print(f"{names[id(p)]:70s} {e1:10.2e} {e2:10.2e}{flag}")
