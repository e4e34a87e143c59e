#!/usr/bin/env python3
"""Per-wave clock stamps of the row-panel GEMM (gemm_panel_kernel) on the FFN up-projection shape."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
N, K, M = 64000, 256, 1024
x = torch.randn(N, K, device="cuda").bfloat16(); w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda"); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y)
e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)
fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
lib = L.lib(); lib.smx_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): fn()
buf = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device="cuda")
lib.smx_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.smx_debug_set_timing_buffer(None)
s = buf.view(-1, 8).cpu().double()
s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
t0 = s[:, 0].min()
names = ["prologue (A panel, B0)", "tile 0 (whole)", "tile 1 main loop", "tile 1 settle", "tile 1 phase 0", "tile 1 phase 1", "tiles 2.. (rest)"]
d = s[:, 1:8] - s[:, 0:7]
print(f"waves {len(s)}  kernel span {float(s[:, 7].max() - t0):.0f} ticks  mean wave lifetime {float((s[:, 7] - s[:, 0]).mean()):.0f}")
for i, n in enumerate(names): print(f"  {n:26s} mean {float(d[:, i].mean()):10.0f}  p90 {float(d[:, i].quantile(0.9)):10.0f}")
print(f"  wave start offset: mean {float((s[:, 0] - t0).mean()):.0f} max {float((s[:, 0] - t0).max()):.0f}")
