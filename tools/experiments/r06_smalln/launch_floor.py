#!/usr/bin/env python3
"""What does ONE dependent kernel cost inside a replayed hipGraph on this box?  Chains of (a) a one-element torch add, (b) the smallest
libsmx row kernel (a cast of 256 elements), (c) a 3750 x 512 LayerNorm, (d) the same LayerNorm at 500 x 256."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops
from tools.graph_timer import graph_us
x1 = torch.zeros(1, device="cuda")
print(f"torch add_ on 1 element:          {graph_us(lambda: x1.add_(1.0), reps=200):5.2f} us per launch")
xs = torch.zeros(256, device="cuda")
print(f"smx cast 256 elements:            {graph_us(lambda: ops.cast(xs, torch.bfloat16), reps=200):5.2f} us per launch")
for N, D in ((3750, 512), (500, 256), (64, 256)):
    x = torch.randn(N, D, device="cuda"); g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    print(f"layernorm_fwd {N} x {D} fp32->bf16: {graph_us(lambda: ops.layernorm_fwd(x, g, b, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16), reps=200):5.2f} us per launch")
