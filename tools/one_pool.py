#!/usr/bin/env python3
"""Run the config-5 pool kernel a few times (profiling target)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import ops
B, T, D = 8, 30000, 512
import time
secs = 0.0
try:
    secs = float(sys.argv[1])                # a number: loop for that many seconds (tools/power_trace.sh), bf16
except (IndexError, ValueError):
    pass
dtype = torch.bfloat16 if (len(sys.argv) < 2 or secs > 0 or sys.argv[1] == "bf16") else torch.float32
s = torch.randn(B * T, D, device="cuda").to(dtype)
lens = torch.randint(T // 2, T + 1, (B,), device="cuda"); lens[0] = T
mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).reshape(-1).view(torch.uint8)
n, t0 = 0, time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while n < 20 or time.time() - t0 < secs:
    for _ in range(20):
        ops.masked_mean(s, mask, B, T, True, False)
    n += 20
    torch.cuda.synchronize()
e1.record()
e1.synchronize()
if secs > 0:
    print(f"masked_mean ({B},{T},{D}) {e0.elapsed_time(e1) * 1e3 / n:8.1f} us per launch over {n} launches")
