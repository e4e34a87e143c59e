#!/usr/bin/env python
"""Panel-resident GEMM (smx_gemm_panel) against the tiled smx_gemm on the two output-bound FFN launches, N = 64000 frames:
   D=256 F=1024 python tools/panel_bench.py     (C2b)        D=512 F=2048 ... (C2a / C5)        D=512 F=3072 (C4)
Prints us per launch (warmed medians), the algorithmic HBM rate, and the weight-pack cost."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel                                  # noqa: E402
from summarymixing_amd import _lib as L, ops                    # noqa: E402

N, d, f = int(os.environ.get("N", 64000)), int(os.environ.get("D", 256)), int(os.environ.get("F", 1024))
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x = rnd(N, d).bfloat16()
W1, W2 = (rnd(f, d) * 0.06).bfloat16(), (rnd(d, f) * 0.03).bfloat16()
b1 = rnd(f) * 0.1
dy = rnd(N, d).bfloat16()
z = rnd(N, f).bfloat16()


def T(fn):
    return sorted(time_kernel(fn, iters=30, warm=8) * 1e6 for _ in range(3))[1]


_wa, _wb = torch.randn(8192, 8192, device=dev).bfloat16(), torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(60):
    torch.matmul(_wa, _wb)
del _wa, _wb
zb, ab = torch.empty(N, f, device=dev, dtype=torch.bfloat16), torch.empty(N, f, device=dev, dtype=torch.bfloat16)
wp1, wp2 = ops.weight_pack(W1, bias=b1), ops.weight_pack(W2, transposed=True)
print(f"# panel vs tiled, N = {N}, d = {d}, d_ffn = {f}, bf16, dropout 0.15 (us; GB/s = algorithmic bytes / time)")
for name, drop in (("", None), (" + dropout", (0.15, 7))):
    nb = (N * d + f * d) * 2 + 2 * N * f * 2
    t0 = T(lambda: ops.gemm(L.GEMM_NT, x, W1, ab, N, f, d, ops.epilogue(bias=b1, act=L.ACT_SWISH, z=zb, drop=drop)))
    t1 = T(lambda: ops.gemm_panel(x, wp1, ab, N, f, d, ops.epilogue(act=L.ACT_SWISH, z=zb, drop=drop)))
    print(f"  up-projection + bias + Swish + Z{name:12s} tiled {t0:7.1f}  panel {t1:7.1f}  ratio {t1 / t0:5.2f}   {nb / t1 * 1e-3:7.0f} GB/s")
    t0 = T(lambda: ops.gemm(L.GEMM_NN, dy, W2, ab, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=drop)))
    t1 = T(lambda: ops.gemm_panel(dy, wp2, ab, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=drop)))
    print(f"  act-grad dgrad (Swish'){name:21s} tiled {t0:7.1f}  panel {t1:7.1f}  ratio {t1 / t0:5.2f}   {nb / t1 * 1e-3:7.0f} GB/s")
t0 = T(lambda: ops.gemm(L.GEMM_NT, x, W1, ab, N, f, d, ops.epilogue(bias=b1)))
t1 = T(lambda: ops.gemm_panel(x, wp1, ab, N, f, d, ops.epilogue()))
print(f"  bias only (one output)                       tiled {t0:7.1f}  panel {t1:7.1f}  ratio {t1 / t0:5.2f}")
print(f"  weight_pack ({f}x{d}): {T(lambda: ops.weight_pack(W1, bias=b1, out=wp1)):.1f} us, transposed: {T(lambda: ops.weight_pack(W2, transposed=True, out=wp2)):.1f} us")
