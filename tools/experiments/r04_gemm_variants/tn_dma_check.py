#!/usr/bin/env python3
"""wgrad through the LDS-DMA TN kernel (SMX_TN_DMA=1) vs fp32 torch: dW += dZ^T X and the bias gradient; timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
torch.manual_seed(0)
mode = os.environ.get("SMX_TN_DMA", "default")
for (N, M, K) in ((64000, 1024, 256), (64000, 256, 1024), (64000, 256, 256), (64000, 512, 256), (33024, 384, 512), (64000, 3072, 512)):
    dz = torch.randn(N, M, device="cuda").bfloat16()
    x = torch.randn(N, K, device="cuda").bfloat16()
    gw = torch.zeros(M, K, device="cuda"); gb = torch.zeros(M, device="cuda")
    ops.wgrad(dz, x, gw, N, M, K, dbias=gb)
    ref = dz.float().t() @ x.float()
    refb = dz.float().sum(0)
    ew = (gw - ref).abs().max().item() / ref.abs().max().item()
    eb = (gb - refb).abs().max().item() / refb.abs().max().item()
    t = time_kernel(lambda: ops.wgrad(dz, x, gw, N, M, K, dbias=gb), 20, 3)
    print(f"TN_DMA={mode} N={N} M={M} K={K}: {t*1e6:7.1f} us  {2.0*N*M*K/t/1e12:6.1f} TF/s  err dW {ew:.1e} db {eb:.1e}", flush=True)
