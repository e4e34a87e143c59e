#!/usr/bin/env python3
"""Run the Conformer conv module's depthwise conv (GLU, k = 31) forward / backward at (128, 500, 256) a few times: PMC target."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
B, T, D, k = 128, 500, 256, 31
p = torch.randn(B * T, 2 * D, device="cuda").bfloat16()
w = torch.randn(D, k, device="cuda") * 0.1; bias = torch.randn(D, device="cuda")
dy = torch.randn(B * T, D, device="cuda").bfloat16()
dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
for _ in range(10):
    ops.dwconv_fwd(p, w, bias, B, T, D, k, True, L.PAD_ZERO, 0)
    ops.dwconv_bwd(dy, p, w, bias, dw, db, B, T, D, k, True, L.PAD_ZERO, 0)
torch.cuda.synchronize()
