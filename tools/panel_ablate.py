#!/usr/bin/env python3
"""FFN up-projection shape through the panel / tiled GEMM with SMX_GEMM_ABLATE (1: no epilogue, 2: no MFMA)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
N, K, M = 64000, 256, 1024
x = torch.randn(N, K, device="cuda").bfloat16(); w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda"); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y)
for name, e in (("bias+swish+Z", ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)), ("bias", ops.epilogue(bias=b))):
    t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e), 20, 3)
    print(f"PANEL={os.environ.get('SMX_GEMM_PANEL','1')} ABLATE={os.environ.get('SMX_GEMM_ABLATE','0')} {name:14s} {t*1e6:7.1f} us", flush=True)
