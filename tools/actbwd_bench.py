#!/usr/bin/env python3
"""act_mask_bwd (+bias column sums) micro-benchmark at the step's shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
from bench import time_kernel
N = int(os.environ.get("N", 64000))
for M in (256, 512, 1024):
    dy = torch.randn(N, M, device="cuda").bfloat16(); z = torch.randn(N, M, device="cuda").bfloat16()
    dz = torch.empty_like(dy); gb = torch.zeros(M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.2).view(torch.uint8)
    t1 = time_kernel(lambda: ops.act_mask_bwd(dy, z, mask, L.ACT_SWISH, 1.0, dz, gb), iters=20, warm=3)
    t2 = time_kernel(lambda: ops.act_mask_bwd(dy, None, None, L.ACT_NONE, 0.5, dz, gb, drop=(0.15, 99)), iters=20, warm=3)
    t3 = time_kernel(lambda: ops.act_mask_bwd(dy, None, None, L.ACT_NONE, 1.0, None, gb), iters=20, warm=3)
    print(f"M={M:5d}: act+mask+bias {t1*1e6:6.1f} us ({3*N*M*2/t1/1e9:5.0f} GB/s) | alpha+drop+bias {t2*1e6:6.1f} us | bias only {t3*1e6:6.1f} us ({N*M*2/t3/1e9:5.0f} GB/s)", flush=True)
