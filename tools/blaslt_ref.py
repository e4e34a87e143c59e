#!/usr/bin/env python3
"""Reference point only (not used by the product): torch.matmul / F.linear (hipBLASLt) on the step's GEMM shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
N = int(os.environ.get("N", 64000))
for (K, M) in ((256, 1024), (1024, 256), (256, 512), (256, 256)):
    x = torch.randn(N, K, device="cuda").bfloat16(); w = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(M, device="cuda").bfloat16()
    t = time_kernel(lambda: torch.nn.functional.linear(x, w, b), iters=20, warm=3)
    print(f"F.linear NT N={N} K={K} M={M}: {t*1e6:7.1f} us  {(N*K+N*M+M*K)*2/t/1e9:6.0f} GB/s")
    dz = torch.randn(N, M, device="cuda").bfloat16()
    t = time_kernel(lambda: torch.matmul(dz, w), iters=20, warm=3)
    print(f"matmul  NN (N,{M})x({M},{K}): {t*1e6:7.1f} us  {(N*K+N*M+M*K)*2/t/1e9:6.0f} GB/s")
    t = time_kernel(lambda: torch.matmul(dz.t(), x), iters=20, warm=3)
    print(f"matmul  TN ({M},N)x(N,{K}): {t*1e6:7.1f} us  {(N*K+N*M)*2/t/1e9:6.0f} GB/s")
