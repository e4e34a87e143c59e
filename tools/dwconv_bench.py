#!/usr/bin/env python3
"""Conformer conv module depthwise conv (GLU, zero pad, k=31) forward / backward at the C2b / C2a widths."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
k = 31
for (B, T, D) in ((128, 500, 256), (128, 500, 512)):
    p = torch.randn(B * T, 2 * D, device="cuda").bfloat16()
    w = torch.randn(D, k, device="cuda") * 0.1; bias = torch.randn(D, device="cuda")
    dy = torch.randn(B * T, D, device="cuda").bfloat16()
    dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
    tf = time_kernel(lambda: ops.dwconv_fwd(p, w, bias, B, T, D, k, True, L.PAD_ZERO, 0), 20, 3)
    tb = time_kernel(lambda: ops.dwconv_bwd(dy, p, w, bias, dw, db, B, T, D, k, True, L.PAD_ZERO, 0), 20, 3)
    n = B * T * D * 2
    print(f"B={B} T={T} D={D}: fwd {tf*1e6:6.1f} us ({3*n/tf/1e9:5.0f} GB/s)  bwd {tb*1e6:6.1f} us ({5*n/tb/1e9:5.0f} GB/s)", flush=True)
