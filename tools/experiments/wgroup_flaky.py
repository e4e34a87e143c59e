#!/usr/bin/env python3
"""Does smx_wgrad_group give bit-identical results when OTHER kernels ran in between (stale LDS contents)?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summarymixing_amd import functional as F, ops, _lib as L
from tools.gemm_bench import run
torch.manual_seed(4)
rows, M, K = 20000, 512, 256
dz = (torch.randn(rows, M, device="cuda") * 0.5).bfloat16()
x = torch.randn(rows, K, device="cuda").bfloat16()
ref = None
bad = 0
for it in range(40):
    if it % 2 == 1:
        run(4096, 1024, 256, "NTln")          # dirty the LDS with another kernel's operand ring / staging rows
        junk = torch.full((1 << 20,), float("nan"), device="cuda")
        del junk
    gW, gb = torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")
    F._wgrad(dz, x, gW, rows, M, K, gb)
    F.flush_deferred()
    torch.cuda.synchronize()
    if ref is None:
        ref = (gW.clone(), gb.clone()); continue
    dW = (gW - ref[0]).abs(); db = (gb - ref[1]).abs()
    nan = int(torch.isnan(gW).sum()) + int(torch.isnan(gb).sum())
    if nan or dW.max() > 0 or db.max() > 0:
        bad += 1
        idx = (dW > 0).nonzero()
        print(it, "nan", nan, "dW diff n=", int((dW > 0).sum()), "max", float(dW.max()), "rows", idx[:, 0].unique()[:8].tolist(),
              "cols", idx[:, 1].unique()[:8].tolist(), "db n=", int((db > 0).sum()), flush=True)
print("bad", bad, "of 39")
