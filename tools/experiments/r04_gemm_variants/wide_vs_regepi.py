#!/usr/bin/env python3
"""Plain dgrad-shaped (NN) and bias+act (NT) GEMMs: wide 128 x 256 tile vs 128 x 128 tile with the register-domain epilogue
(run with SMX_GEMM_WIDE=0/1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
N = 64000
for (K, M) in ((1024, 256), (512, 256), (256, 256), (2048, 512), (512, 512)):
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16(); wt = w.t().contiguous()
    b = torch.randn(M, device="cuda")
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    t1 = time_kernel(lambda: ops.gemm(L.GEMM_NN, x, wt, y, N, M, K), 20, 3)
    ep = ops.epilogue(bias=b, act=L.ACT_SWISH)
    t2 = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ep), 20, 3)
    print(f"WIDE={os.environ.get('SMX_GEMM_WIDE','auto')} REG_EPI={os.environ.get('SMX_REG_EPI','default')} K={K} M={M}: NN plain {t1*1e6:6.1f} us | NT bias+swish {t2*1e6:6.1f} us", flush=True)
