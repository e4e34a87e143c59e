#!/usr/bin/env python3
"""Does the relative placement of the Y and Z outputs of the FFN up-projection matter once they no longer fit the MALL?
usage: zoff_probe.py [N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
from bench import time_kernel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
K, M = 256, 1024
x = torch.randn(N, K, device="cuda").bfloat16()
w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda")
y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH)), iters=20, warm=3)
print(f"N={N} Y only                : {t*1e6:7.1f} us")
pool = torch.empty(N * M + (64 << 20), device="cuda", dtype=torch.bfloat16)
for off in (0, 512, 2048, 8192, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20):
    z = pool[off:off + N * M].view(N, M)
    t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)), iters=20, warm=3)
    d = (z.data_ptr() - y.data_ptr())
    print(f"N={N} Y+Z  z-y = {d:>12d} B (mod 2MiB {d % (2 << 20):>8d}, off {off*2:>8d} B): {t*1e6:7.1f} us")
# one interleaved buffer: Y = cols [0,M), Z = cols [M,2M) of an (N, 2M) matrix
yz = torch.empty(N, 2 * M, device="cuda", dtype=torch.bfloat16)
t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, yz[:, :M], N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=yz[:, M:])), iters=20, warm=3)
print(f"N={N} Y|Z interleaved rows  : {t*1e6:7.1f} us")
