#!/usr/bin/env python3
"""A few launches of one smx_gemm_panel configuration (for rocprofv3 --pmc passes): one_panel.py fwd|ag N K M"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
mode, N, K, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(N, K, device="cuda").bfloat16()
W = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda")
y, z = torch.empty(N, M, device="cuda", dtype=torch.bfloat16), torch.randn(N, M, device="cuda").bfloat16()
wp = ops.weight_pack(W, bias=b if mode == "fwd" else None)
e = ops.epilogue(act=L.ACT_SWISH, z=z, drop=(0.15, 7)) if mode == "fwd" else ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 7))
for _ in range(6):
    ops.gemm_panel(x, wp, y, N, M, K, e)
torch.cuda.synchronize()
