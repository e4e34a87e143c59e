#!/usr/bin/env python3
"""The small row-wise kernels of a C2b layer in isolation (back to back, hot caches): where they stand against the bytes they move."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
B, T, D = 128, 500, 256
N = B * T
bf = torch.bfloat16
cat = torch.randn(N, 2 * D, device="cuda").to(bf)
sbar = torch.randn(B, D, device="cuda")
inv = torch.rand(B, device="cuda") + 0.5
z = torch.randn(N, 2 * D, device="cuda").to(bf)
mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
x = torch.randn(N, D, device="cuda").to(bf)
gam, bet = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
def rep(name, fn, nbytes):
    t = time_kernel(fn, 50, 5)
    print(f"{name:44s} {t*1e6:7.1f} us  {nbytes/t/1e9:6.0f} GB/s", flush=True)
rep("bcast_rows + dropout -> cat[:, D:] (fwd)", lambda: ops.bcast_rows(sbar, None, cat[:, D:], B, T, drop=(0.15, 77)), N * D * 2)
rep("bcast_rows plain -> cat[:, D:]", lambda: ops.bcast_rows(sbar, None, cat[:, D:], B, T), N * D * 2)
rep("bcast_rows + act_bwd(z, mask) (bwd)", lambda: ops.bcast_rows_act_bwd(sbar, inv, cat[:, D:], B, T, z[:, D:], mask, L.ACT_SWISH), 2 * N * D * 2)
rep("masked_mean pool of cat[:, D:]", lambda: ops.masked_mean(cat[:, D:], mask, B, T, want_inv=True), N * D * 2)
rep("masked_mean (sum, no mask) contiguous", lambda: ops.masked_mean(x, None, B, T, scale=False), N * D * 2)
rep("layernorm_fwd (+stats)", lambda: ops.layernorm_fwd(x, gam, bet, 1e-5, True), 2 * N * D * 2)
rep("layernorm_fwd + swish", lambda: ops.layernorm_fwd(x, gam, bet, 1e-5, True, L.ACT_SWISH), 2 * N * D * 2)
y, st = ops.layernorm_fwd(x, gam, bet, 1e-5, True)
dy = torch.randn(N, D, device="cuda").to(bf)
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
rep("layernorm_bwd (+res)", lambda: ops.layernorm_bwd(dy, x, gam, bet, st, dg, db, res=dy), 4 * N * D * 2)
rep("layernorm_bwd (+res, second out)", lambda: ops.layernorm_bwd(dy, x, gam, bet, st, dg, db, res=dy, second=(0.5, None, (0.15, 5))), 5 * N * D * 2)
