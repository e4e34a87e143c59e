#!/usr/bin/env python3
"""Per-wave s_memtime stamps of one smx_gemm launch: where does a wave's lifetime go?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
N, K, M = 32000, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
x = torch.randn(N, K, device="cuda").bfloat16(); w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y); b = torch.randn(M, device="cuda")
e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)
lib = L.lib(); lib.smx_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
nblk = ((N + 127) // 128) * ((M + 127) // 128)
buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
lib.smx_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e); torch.cuda.synchronize()
lib.smx_debug_set_timing_buffer(None)
s = buf.view(nblk * 4, 8).cpu().double()
t0 = s[:, 0].min()
names = ["start->loads issued", "main loop", "stage ph0 (+barriers)", "loop ph0", "stage ph1", "loop ph1", "tail"]
d = s[:, 1:8] - s[:, 0:7]
print("kernel span (clock ticks):", float(s[:, 7].max() - t0), " mean wave lifetime:", float((s[:, 7] - s[:, 0]).mean()))
for i, n in enumerate(names): print(f"  {n:26s} mean {float(d[:, i].mean()):10.0f}  p90 {float(d[:, i].quantile(0.9)):10.0f}")
print("  wave start offset: mean", float((s[:, 0] - t0).mean()), " max", float((s[:, 0] - t0).max()))
