#!/usr/bin/env python3
"""Row-panel GEMM kernel (gemm_panel_kernel: K <= 256, M >= 512) vs the tiled kernel: correctness against torch and
timing on the output-heavy shapes of the step.  SMX_GEMM_PANEL=0/1/2 selects the path (read once per process)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
torch.manual_seed(0)
mode = os.environ.get("SMX_GEMM_PANEL", "1")
for (N, K, M, what) in ((64000, 256, 1024, "ffn-up"), (64000, 256, 512, "global_proj"), (64000, 192, 520, "ragged"),
                        (32768 + 77, 64, 640, "k64"), (64000, 512 if False else 256, 2048, "wide")):
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); z = torch.empty_like(y)
    e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, row_mask=mask)
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
    zr = x.float() @ w.float().t() + b
    yr = torch.nn.functional.silu(zr) * mask[:, None]
    ez = (z.float() - zr).abs().max().item() / zr.abs().max().item()
    ey = (y.float() - yr).abs().max().item() / yr.abs().max().item()
    t = time_kernel(lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e), 20, 3)
    nb = (N * K + M * K + 2 * N * M) * 2
    # NN layout (dgrad-shaped): B is (K, M)
    wt = w.t().contiguous()
    y2 = torch.empty_like(y)
    ops.gemm(L.GEMM_NN, x, wt, y2, N, M, K, ops.epilogue(bias=b))
    e2 = (y2.float() - zr).abs().max().item() / zr.abs().max().item()
    t2 = time_kernel(lambda: ops.gemm(L.GEMM_NN, x, wt, y2, N, M, K, ops.epilogue(bias=b)), 20, 3)
    print(f"PANEL={mode} {what:12s} ({N}x{K})x({K}x{M}): NT+bias+swish+Z+mask {t*1e6:7.1f} us ({nb/t/1e12:4.2f} TB/s) err z {ez:.1e} y {ey:.1e}"
          f" | NN+bias {t2*1e6:7.1f} us err {e2:.1e}", flush=True)
