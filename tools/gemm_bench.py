#!/usr/bin/env python3
"""Micro-benchmark of the smx_gemm layouts on the shapes of the C2b/C2a training step (HIP-event timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops  # noqa: E402
from bench import time_kernel  # noqa: E402


def run(N, K, M, layout, dtype=torch.bfloat16, epi="swishz"):
    x = torch.randn(N, K, device="cuda").to(dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    if layout == "NT":
        w = (torch.randn(M, K, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=dtype)
        z = torch.empty(N, M, device="cuda", dtype=dtype)
        b = torch.randn(M, device="cuda")
        e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z) if epi == "swishz" else ops.epilogue(bias=b)
        fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K + (2 if epi == "swishz" else 1) * N * M) * es
    elif layout == "NNag":  # dgrad with the fused activation backward: dZ_up (N,M) = (dZ (N,K) W (K,M)) * act'(Z_up), Z read
        w = (torch.randn(K, M, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=dtype)
        zin = torch.randn(N, M, device="cuda").to(dtype)
        e = ops.epilogue(act=L.ACT_SWISH, act_grad_z=zin, drop=(0.15, 1234))
        fn = lambda: ops.gemm(L.GEMM_NN, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K + 2 * N * M) * es
    elif layout == "NTres":  # forward Linear with residual + dropout + alpha (FFN down-projection)
        w = (torch.randn(M, K, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=dtype)
        r = torch.randn(N, M, device="cuda").to(dtype)
        b = torch.randn(M, device="cuda")
        e = ops.epilogue(bias=b, res=r, alpha=0.5, drop=(0.15, 99))
        fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K + 2 * N * M) * es
    elif layout == "NN":    # dgrad: dX (N,M) = dZ (N,K) W (K,M)
        w = (torch.randn(K, M, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=dtype)
        fn = lambda: ops.gemm(L.GEMM_NN, x, w, y, N, M, K)
        nbytes = (N * K + M * K + N * M) * es
    else:                   # wgrad: dW (K,M) += dZ^T (N,K)^T X (N,M): reduce over N
        x2 = torch.randn(N, M, device="cuda").to(dtype)
        g = torch.zeros(K, M, device="cuda")
        fn = lambda: ops.wgrad(x, x2, g, N, K, M)
        nbytes = (N * K + N * M) * es + K * M * 4
    t = time_kernel(fn, iters=20, warm=3)
    fl = 2.0 * N * K * M
    print(f"{layout} N={N:6d} K={K:5d} M={M:5d} {epi:7s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  {nbytes/t/1e9:7.0f} GB/s(alg)", flush=True)


if __name__ == "__main__":
    N = int(os.environ.get("N", 32000))
    for d in (256, 512):
        f = 4 * d
        run(N, d, f, "NT"); run(N, f, d, "NT", epi="plain"); run(N, d, 2 * d, "NT"); run(N, d, d, "NT", epi="plain")
        run(N, f, d, "NN"); run(N, d, f, "NN"); run(N, d, d, "NN")
        run(N, f, d, "TN"); run(N, d, f, "TN"); run(N, d, d, "TN")
        run(N, d, f, "NNag"); run(N, f, d, "NTres")
