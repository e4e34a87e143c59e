#!/usr/bin/env python3
"""O(T) SummaryMixing-expdecay summary at the long-utterance point (8, 30000, 512): the dense (T,T) path of the
reference would need a 3.6 GB fp32 matrix and 2*T*D = 30.7 MFLOP per frame."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import ops
from bench import time_kernel
for (B, T, D) in ((8, 30000, 512), (128, 500, 256)):
    for dtype in (torch.bfloat16, torch.float32):
        s = torch.randn(B * T, D, device="cuda").to(dtype)
        out = torch.empty_like(s)
        t = time_kernel(lambda: ops.expdecay_mean(s, out, B, T, 0.995), iters=10, warm=2)
        nb = 3 * B * T * D * s.element_size()          # S read twice (chunk pass + apply pass), out written once
        print(f"expdecay_mean ({B},{T},{D}) {str(dtype):15s}: {t*1e6:8.1f} us  {nb/t/1e9:6.0f} GB/s (3 passes over the tensor)")
