#!/usr/bin/env python3
"""Yardstick only (never a dependency of the product): torch.matmul / F.linear (hipBLASLt) on the step's GEMM shapes.

    python tools/blaslt_ref.py                 # timings (HIP events), d = 256 and d = 512 shapes at N frames
    rocprofv3 --kernel-trace --stats -- python tools/blaslt_ref.py   # + the Tensile kernel names (macro tile etc.)

Data is uniform random in [-1, 1) (a zero fill would flatter the library through DVFS)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
N = int(os.environ.get("N", 64000))
SHAPES = ((256, 1024), (1024, 256), (256, 512), (512, 256), (256, 256), (512, 2048), (2048, 512), (512, 512), (512, 1024))
if len(sys.argv) > 1:
    SHAPES = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
for (K, M) in SHAPES:
    x = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(M, device="cuda").bfloat16()
    t = time_kernel(lambda: torch.nn.functional.linear(x, w, b), iters=20, warm=3)
    print(f"F.linear NT ({N},{K})x({M},{K})^T +bias: {t*1e6:7.1f} us  {(N*K+N*M+M*K)*2/t/1e9:6.0f} GB/s  {2*N*K*M/t/1e12:6.0f} TF")
    dz = (torch.rand(N, M, device="cuda") * 2 - 1).bfloat16()
    t = time_kernel(lambda: torch.matmul(dz, w), iters=20, warm=3)
    print(f"matmul  NN ({N},{M})x({M},{K}):        {t*1e6:7.1f} us  {(N*K+N*M+M*K)*2/t/1e9:6.0f} GB/s  {2*N*K*M/t/1e12:6.0f} TF")
    t = time_kernel(lambda: torch.matmul(dz.t(), x), iters=20, warm=3)
    print(f"matmul  TN ({M},{N})x({N},{K}):        {t*1e6:7.1f} us  {(N*K+N*M)*2/t/1e9:6.0f} GB/s  {2*N*K*M/t/1e12:6.0f} TF")
