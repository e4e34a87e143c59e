#!/usr/bin/env python3
"""HBM bandwidth probes on the box: fill / copy / my axpby, to calibrate what 'peak' means for writes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import ops
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    b = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    t = time_kernel(lambda: a.zero_(), 20, 3); print(f"fill  {mb:5d} MB: {t*1e6:8.1f} us  {mb/1024/t/1e0*1.073741824:8.1f} GB/s written")
    t = time_kernel(lambda: b.copy_(a), 20, 3); print(f"copy  {mb:5d} MB: {t*1e6:8.1f} us  {2*mb/1024/t*1.073741824:8.1f} GB/s r+w")
    a2, b2 = a.view(-1, 1024), b.view(-1, 1024)
    t = time_kernel(lambda: ops.axpby(1.0, a2, out=b2), 20, 3); print(f"axpby {mb:5d} MB: {t*1e6:8.1f} us  {2*mb/1024/t*1.073741824:8.1f} GB/s r+w")
    t = time_kernel(lambda: a.sum(), 20, 3); print(f"read  {mb:5d} MB: {t*1e6:8.1f} us  {mb/1024/t*1.073741824:8.1f} GB/s read (torch sum)")
