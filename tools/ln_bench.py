#!/usr/bin/env python3
"""LayerNorm forward/backward micro-benchmark at the C2b/C2a row sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
from bench import time_kernel

N = int(os.environ.get("N", 64000))
for D in (256, 512, 1024):
    for dtype in (torch.bfloat16, torch.float32):
        x = torch.randn(N, D, device="cuda").to(dtype)
        dy = torch.randn(N, D, device="cuda").to(dtype)
        g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
        dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
        es = x.element_size()
        _, st = ops.layernorm_fwd(x, g, b, 1e-5, True)
        tf = time_kernel(lambda: ops.layernorm_fwd(x, g, b, 1e-5, True), iters=20, warm=3)
        tfa = time_kernel(lambda: ops.layernorm_fwd(x, g, b, 1e-5, True, L.ACT_SWISH), iters=20, warm=3)
        tb = time_kernel(lambda: ops.layernorm_bwd(dy, x, g, b, st, dg, db, dy), iters=20, warm=3)
        print(f"D={D:5d} {str(dtype):15s} fwd {tf*1e6:6.1f} us {2*N*D*es/tf/1e9:6.0f} GB/s | fwd+swish {tfa*1e6:6.1f} us | "
              f"bwd(+res) {tb*1e6:6.1f} us {4*N*D*es/tb/1e9:6.0f} GB/s", flush=True)
