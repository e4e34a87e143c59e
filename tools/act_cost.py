#!/usr/bin/env python3
"""Cost of the activation in a GEMM epilogue: bias only vs Swish vs exact-erf GELU (+ saved Z), Branchformer pre-projection shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
N, K, M = 32000, 512, 3072
x = torch.randn(N, K, device="cuda").bfloat16(); w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda"); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y)
dy = torch.randn(N, M, device="cuda").bfloat16(); dz = torch.empty_like(dy)
for name, act in (("none", L.ACT_NONE), ("swish", L.ACT_SWISH), ("gelu", L.ACT_GELU)):
    t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=act, z=z if act != L.ACT_NONE else None)), 20, 3)
    tb = time_kernel(lambda: ops.act_mask_bwd(dy, z, None, act, 1.0, dz, None), 20, 3) if act != L.ACT_NONE else 0.0
    print(f"act={name:5s}: GEMM fwd {t*1e6:7.1f} us | act_mask_bwd {tb*1e6:7.1f} us", flush=True)
