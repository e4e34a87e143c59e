#!/usr/bin/env python3
"""Run the config-5 pool kernel a few times (profiling target)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import ops
B, T, D = 8, 30000, 512
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
s = torch.randn(B * T, D, device="cuda").to(dtype)
lens = torch.randint(T // 2, T + 1, (B,), device="cuda"); lens[0] = T
mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).reshape(-1).view(torch.uint8)
for _ in range(20):
    ops.masked_mean(s, mask, B, T, True, False)
torch.cuda.synchronize()
