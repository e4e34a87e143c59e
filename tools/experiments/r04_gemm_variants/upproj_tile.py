#!/usr/bin/env python3
"""d=512 up-projection shapes (K=512 -> M=2048 / 1024), bias+swish without Z (eval) and with Z (training): tile choice."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
N = 64000
for (K, M) in ((512, 2048), (512, 1024), (256, 1024), (256, 512)):
    x = torch.randn(N, K, device="cuda").bfloat16(); w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(M, device="cuda"); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y)
    t1 = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH)), 20, 3)
    t2 = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)), 20, 3)
    print(f"WIDE={os.environ.get('SMX_GEMM_WIDE','auto')} K={K} M={M}: bias+swish {t1*1e6:6.1f} us ({2.0*N*K*M/t1/1e12:5.0f} TF/s) | +Z {t2*1e6:6.1f} us", flush=True)
