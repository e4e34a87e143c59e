#!/usr/bin/env python3
"""Micro-benchmark of the smx_gemm layouts on the shapes of the C2b/C2a training step (HIP-event timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops  # noqa: E402
from bench import time_kernel  # noqa: E402


def build(N, K, M, layout, dtype=torch.bfloat16, epi="swishz"):
    """-> (launch closure, algorithmic bytes per launch) of one smx_gemm shape + epilogue."""
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
    elif layout == "NTres32":  # forward Linear + bias + float32 residual -> float32 stream tensor (eval: no dropout; the unfused down-projection)
        w = (torch.randn(M, K, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=torch.float32)
        r = torch.randn(N, M, device="cuda")
        b = torch.randn(M, device="cuda")
        e = ops.epilogue(bias=b, res=r, alpha=0.5, out_mode=L.OUT_F32)
        fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K) * es + N * M * 8
    elif layout in ("NTln", "NTln2"):   # Linear + bias + dropout + alpha + float32 residual -> float32 stream tensor, LayerNorm appended
        # (FFN down-projection K = 4d / merge, conv out-projection K = d of a Conformer layer on the float32 residual stream;
        #  NTln2: the LayerNorm output is the stream itself - the layer-final norm2 - and float32 as well)
        w = (torch.randn(M, K, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=torch.float32)
        r = torch.randn(N, M, device="cuda")
        b = torch.randn(M, device="cuda")
        g_, b_ = torch.ones(M, device="cuda"), torch.zeros(M, device="cuda")
        hy = torch.empty(N, M, device="cuda", dtype=torch.float32 if layout == "NTln2" else dtype)
        st = torch.empty(N, 2, device="cuda")
        e = ops.epilogue(bias=b, res=r, alpha=0.5, drop=(0.15, 99), out_mode=L.OUT_F32, ln_fwd=(g_, b_, hy, st, 1e-5, L.ACT_NONE))
        fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K) * es + N * M * (4 + 4 + hy.element_size())
    elif layout in ("NTlnm", "NTlnc"):   # the cell's merge (K = l + s, +act+Z) / the conv module's out-projection (+mask+drop), LayerNorm appended
        w = (torch.randn(M, K, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=torch.float32)
        r = torch.randn(N, M, device="cuda")
        b = torch.randn(M, device="cuda")
        g_, b_ = torch.ones(M, device="cuda"), torch.zeros(M, device="cuda")
        hy = torch.empty(N, M, device="cuda", dtype=dtype)
        st = torch.empty(N, 2, device="cuda")
        if layout == "NTlnm":
            z = torch.empty(N, M, device="cuda", dtype=dtype)
            e = ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, res=r, out_mode=L.OUT_F32, ln_fwd=(g_, b_, hy, st, 1e-5, L.ACT_NONE))
        else:
            mk = (torch.rand(N, device="cuda") < 0.75).view(torch.uint8)
            e = ops.epilogue(bias=b, res=r, row_mask=mk, drop=(0.15, 99), out_mode=L.OUT_F32, ln_fwd=(g_, b_, hy, st, 1e-5, L.ACT_NONE))
        fn = lambda: ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K) * es + N * M * (4 + 4 + es + (es if layout == "NTlnm" else 0))
    elif layout in ("NNlnb", "NNlnb3"):  # dgrad + LayerNorm backward in the epilogue (float32 LN input, residual gradient, second output)
        w = (torch.randn(K, M, device="cuda") * 0.05).to(dtype)
        y = torch.empty(N, M, device="cuda", dtype=dtype)
        lx = torch.randn(N, M, device="cuda")
        st = torch.stack([lx.mean(1), lx.var(1, unbiased=False).add(1e-5).rsqrt()], 1).contiguous()
        g_, b_ = torch.ones(M, device="cuda"), torch.zeros(M, device="cuda")
        rg = torch.randn(N, M, device="cuda").to(dtype)
        dx2 = torch.empty(N, M, device="cuda", dtype=dtype)
        tr = L.lib().smx_gemm_ln_tile_rows_for(N, M)
        ws = torch.empty(((N + tr - 1) // tr) * 2 * M, device="cuda")
        if layout == "NNlnb3":   # + the consumer's activation gradient in the second output (the cell's dy * act'(z_m))
            z2 = torch.randn(N, M, device="cuda").to(dtype)
            e = ops.epilogue(res=rg, ln_bwd=(lx, st, g_, ws, dx2, (1.0, None, None, z2, L.ACT_SWISH), None, True))
        else:
            e = ops.epilogue(res=rg, ln_bwd=(lx, st, g_, ws, dx2, (0.5, None, (0.15, 7)), None, True))
        fn = lambda: ops.gemm(L.GEMM_NN, x, w, y, N, M, K, e)
        nbytes = (N * K + M * K) * es + N * M * (4 + 3 * es + (es if layout == "NNlnb3" else 0))
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
    return fn, nbytes


def run(N, K, M, layout, dtype=torch.bfloat16, epi="swishz"):
    fn, nbytes = build(N, K, M, layout, dtype, epi)
    ops.prof_start()                                     # the name this launch carries in bench.py's in-step records
    fn()
    rec = ops.prof_stop()
    if rec:   # name AND algorithmic bytes of this launch as bench.py's in-step records carry them (ONE byte model: ops.gemm)
        open("/tmp/pmc_name.txt", "w").write(f"{rec[0][0]}\n{rec[0][1]:.0f}\n")
    t = sorted(time_kernel(fn, iters=30, warm=8) for _ in range(3))[1]   # median of three warmed runs (one cold run of 20 read 5-15 % high)
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
