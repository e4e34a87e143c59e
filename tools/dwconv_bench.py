#!/usr/bin/env python3
"""GLU + depthwise conv (k=31) forward / backward timing vs batch size."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
D, T, k = 256, 500, 31
for B in (32, 64, 96, 128, 256):
    p = torch.randn(B * T, 2 * D, device="cuda").bfloat16()
    w = torch.randn(D, k, device="cuda") * 0.1; bias = torch.randn(D, device="cuda")
    dy = torch.randn(B * T, D, device="cuda").bfloat16()
    dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
    tf = time_kernel(lambda: ops.dwconv_fwd(p, w, bias, B, T, D, k, True), 20, 3)
    tb = time_kernel(lambda: ops.dwconv_bwd(dy, p, w, bias, dw, db, B, T, D, k, True), 20, 3)
    nb_f = B * T * D * 2 * 3; nb_b = B * T * D * 2 * 7
    print(f"B={B:4d}: fwd {tf*1e6:7.1f} us ({nb_f/tf/1e9:6.0f} GB/s)   bwd {tb*1e6:7.1f} us ({nb_b/tb/1e9:6.0f} GB/s)", flush=True)
