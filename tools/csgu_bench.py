#!/usr/bin/env python3
"""CSGU (gated, reflect-padded depthwise conv, k=31, D=1536: the Branchformer cgMLP) forward / backward timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
D, k = 1536, 31
for (B, T) in ((128, 250), (16, 2000), (32, 250)):
    x2 = torch.randn(B * T, D, device="cuda").bfloat16(); gate = torch.randn(B * T, D, device="cuda").bfloat16()
    w = torch.randn(D, k, device="cuda") * 0.1; bias = torch.randn(D, device="cuda")
    dy = torch.randn(B * T, D, device="cuda").bfloat16()
    dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
    for pad, name in ((L.PAD_REFLECT, "reflect"), (L.PAD_ZERO, "zero   ")):
        tf = time_kernel(lambda: ops.dwconv_fwd(x2, w, bias, B, T, D, k, False, pad, 0, gate), 10, 2)
        tb = time_kernel(lambda: ops.dwconv_bwd(dy, x2, w, bias, dw, db, B, T, D, k, False, pad, 0, gate), 10, 2)
        n = B * T * D * 2
        print(f"B={B:4d} T={T:5d} {name}: fwd {tf*1e6:7.1f} us ({3*n/tf/1e9:5.0f} GB/s)   bwd {tb*1e6:7.1f} us ({5*n/tb/1e9:5.0f} GB/s)", flush=True)
