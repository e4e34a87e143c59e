#!/usr/bin/env python3
"""Sensitivity of the 128 x 512 LayerNorm-forward launch to its epilogue parts (64 000 x 2048 -> 512, float32 stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summarymixing_amd import _lib as L, ops
from bench import time_kernel
N, K, M = 64000, int(os.environ.get("K", 2048)), 512
x = torch.randn(N, K, device="cuda").bfloat16()
w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
y = torch.empty(N, M, device="cuda"); yb = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
r = torch.randn(N, M, device="cuda"); rb = r.bfloat16()
b = torch.randn(M, device="cuda"); g_, b_ = torch.ones(M, device="cuda"), torch.zeros(M, device="cuda")
hy = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); st = torch.empty(N, 2, device="cuda")
ln = (g_, b_, hy, st, 1e-5, L.ACT_NONE)
cfgs = [("LN + bias + res32 + drop      ", ops.epilogue(bias=b, res=r, alpha=0.5, drop=(0.15, 99), out_mode=L.OUT_F32, ln_fwd=ln), y),
        ("LN + bias + res32 (no drop)   ", ops.epilogue(bias=b, res=r, alpha=0.5, out_mode=L.OUT_F32, ln_fwd=ln), y),
        ("LN + res32, no bias/alpha/drop", ops.epilogue(res=r, out_mode=L.OUT_F32, ln_fwd=ln), y),
        ("LN + drop, no stats           ", ops.epilogue(bias=b, res=r, alpha=0.5, drop=(0.15, 99), out_mode=L.OUT_F32, ln_fwd=(g_, b_, hy, None, 1e-5, L.ACT_NONE)), y),
        ("no LN: bias + res32 + drop    ", ops.epilogue(bias=b, res=r, alpha=0.5, drop=(0.15, 99), out_mode=L.OUT_F32), y),
        ("no LN: bias only, bf16 out    ", ops.epilogue(bias=b), yb)]
best = [1e9] * len(cfgs)
med = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("ROUNDS", 5))):      # interleaved rounds: min and median per configuration
    for i, (name, e, out) in enumerate(cfgs):
        t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, out, N, M, K, e), iters=30, warm=5) * 1e6
        best[i] = min(best[i], t); med[i].append(t)
print(f"K={K}  (us: min / median of {len(med[0])} interleaved rounds of 30 launches)")
for (name, _, _), b_, m in zip(cfgs, best, med):
    print(name, round(b_, 1), "/", round(sorted(m)[len(m) // 2], 1))
