#!/usr/bin/env python3
"""Per-wave s_memtime stamps of one smx_gemm launch: where does a wave's lifetime go?
usage: gemm_stamps.py [NT|NN|TN] [K] [M]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
layout = sys.argv[1] if len(sys.argv) > 1 else "NT"
N, K, M = int(os.environ.get("N", 32000)), int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 1024
x = torch.randn(N, K, device="cuda").bfloat16()
if layout == "NT":
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y); b = torch.randn(M, device="cuda")
    e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z)
    fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
elif layout == "NNag":     # dgrad with the fused activation backward + dropout (the FFN's dz1 = (dy W2) * act'(z1))
    w = (torch.randn(K, M, device="cuda") * 0.05).bfloat16(); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    zin = torch.randn(N, M, device="cuda").bfloat16()
    e = ops.epilogue(act=L.ACT_SWISH, act_grad_z=zin, drop=(0.15, 1234))
    fn = lambda: ops.gemm(L.GEMM_NN, x, w, y, N, M, K, e)
elif layout == "NTd":      # up-projection with dropout
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y); b = torch.randn(M, device="cuda")
    e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, drop=(0.15, 77))
    fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
elif layout == "NN":
    w = (torch.randn(K, M, device="cuda") * 0.05).bfloat16(); y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm(L.GEMM_NN, x, w, y, N, M, K)
elif layout == "TN":
    x2 = torch.randn(N, M, device="cuda").bfloat16(); g = torch.zeros(K, M, device="cuda")
    fn = lambda: ops.wgrad(x, x2, g, N, K, M)
else:                      # any layout + epilogue of tools/gemm_bench.py (NTln, NTln2, NNlnb, NTres, ...)
    from tools.gemm_bench import build
    fn, _ = build(N, K, M, layout, epi=sys.argv[4] if len(sys.argv) > 4 else "swishz")
assert "diag" in L.LIB_PATH, "run with SMX_LIB=summarymixing_amd/libsmx_diag.so (SMX_DIAG=1 bash summarymixing_amd/csrc/build.sh): the product library carries no stamps"
lib = L.lib(); lib.smx_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): fn()
buf = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device="cuda")
lib.smx_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.smx_debug_set_timing_buffer(None)
s = buf.view(-1, 8).cpu().double()
s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
t0 = s[:, 0].min()
names = ["start->loads issued", "main loop", "stage ph0 (+barriers)", "loop ph0", "stage ph1", "loop ph1", "tail"]
d = s[:, 1:8] - s[:, 0:7]
print(f"{layout} K={K} M={M}: waves {len(s)}  kernel span {float(s[:, 7].max() - t0):.0f} ticks  mean wave lifetime {float((s[:, 7] - s[:, 0]).mean()):.0f}")
for i, n in enumerate(names): print(f"  {n:26s} mean {float(d[:, i].mean()):10.0f}  p90 {float(d[:, i].quantile(0.9)):10.0f}")
print(f"  wave start offset: mean {float((s[:, 0] - t0).mean()):.0f} max {float((s[:, 0] - t0).max()):.0f}")
