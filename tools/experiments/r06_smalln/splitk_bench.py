#!/usr/bin/env python3
"""Split-K slabs + the reducer (smx_gemm_panel_slabs + smx_slab_epilogue) against the tiled smx_gemm + standalone LayerNorm, inside a
replayed hipGraph (tools/graph_timer.py).   N=3750 D=512 python tools/experiments/r06_smalln/splitk_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops
from tools.graph_timer import graph_us

d = int(os.environ.get("D", 512))
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
for N in [int(v) for v in os.environ.get("NS", "500,2000,3750,6000,8000,15000").split(",")]:
    for K in (4 * d, 2 * d, d):
        M, ks = d, d
        ns = K // ks
        x, W, b = rnd(N, K).bfloat16(), (rnd(M, K) * 0.05).bfloat16(), rnd(M) * 0.1
        res = rnd(N, M)
        g1, b1 = torch.ones(M, device=dev), torch.zeros(M, device=dev)
        c, y, st = torch.empty(N, M, device=dev), torch.empty(N, M, device=dev, dtype=torch.bfloat16), torch.empty(N, 2, device=dev)
        slabs = torch.empty(ns, N, M, device=dev)
        wp = ops.weight_pack_slices(W, ks)
        kw = dict(bias=b, res=res, alpha=0.5, drop=(0.15, 9), out_mode=L.OUT_F32)
        t_gemm = graph_us(lambda: ops.gemm(L.GEMM_NT, x, W, c, N, M, K, ops.epilogue(**kw)))
        t_ln = graph_us(lambda: ops.layernorm_fwd(c, g1, b1, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16))
        t_both = graph_us(lambda: (ops.gemm(L.GEMM_NT, x, W, c, N, M, K, ops.epilogue(**kw)), ops.layernorm_fwd(c, g1, b1, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16)), reps=20)
        t_sl = graph_us(lambda: ops.gemm_panel_slabs(x, wp, slabs, N, M, ks, ns))
        t_ep = graph_us(lambda: ops.slab_epilogue(slabs, ns, c, N, M, ops.epilogue(ln_fwd=(g1, b1, y, st, 1e-5, L.ACT_NONE), **kw)))
        t_new = graph_us(lambda: (ops.gemm_panel_slabs(x, wp, slabs, N, M, ks, ns), ops.slab_epilogue(slabs, ns, c, N, M, ops.epilogue(ln_fwd=(g1, b1, y, st, 1e-5, L.ACT_NONE), **kw))), reps=20)
        print(f"d={d} N={N:6d} K={K:5d} S={ns}: tiled gemm {t_gemm:6.1f} + LN {t_ln:5.1f} = pair {t_both:6.1f} | slabs {t_sl:6.1f} + reducer {t_ep:5.1f} = pair {t_new:6.1f}", flush=True)
