#!/usr/bin/env python3
"""Config-5 pool kernel (masked mean over T) bandwidth, bf16 and fp32, full and ragged lengths."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import ops
for dtype in (torch.bfloat16, torch.float32):
    for (B, T, D) in ((8, 30000, 512), (64, 500, 256), (64, 500, 512)):
        s = torch.randn(B * T, D, device="cuda").to(dtype)
        lens = torch.randint(T // 2, T + 1, (B,), device="cuda"); lens[0] = T
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).reshape(-1).view(torch.uint8)
        t = time_kernel(lambda: ops.masked_mean(s, mask, B, T, True, False), 30, 5)
        nb = B * T * D * s.element_size() + B * T + 4 * B * D
        print(f"{str(dtype):16s} ({B},{T},{D}): {t*1e6:7.1f} us  {nb/t/1e9:7.0f} GB/s  ({nb/t/1e9/8000*100:4.1f}% of 8 TB/s)", flush=True)
