#!/usr/bin/env python3
"""Sweep of the panel GEMM's panel height against the tiled kernel over frame counts (one process per SMX_PANEL_ROWS; graph-replay timing).
Prints one line per (N, M, case): us.   D=512 SMX_PANEL_ROWS=64 python tools/experiments/r06_smalln/panel_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops
from tools.graph_timer import graph_us

d = int(os.environ.get("D", 512))
rows = L.get_config()["panel_rows"]
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
for N in [int(v) for v in os.environ.get("NS", "2000,3750,6000,8000,12000,20000").split(",")]:
    x, dy = rnd(N, d).bfloat16(), rnd(N, d).bfloat16()
    for f in (4 * d, 2 * d, d):
        W1, W2 = (rnd(f, d) * 0.06).bfloat16(), (rnd(d, f) * 0.03).bfloat16()
        b1 = rnd(f) * 0.1
        zb, ab = torch.empty(N, f, device=dev, dtype=torch.bfloat16), torch.empty(N, f, device=dev, dtype=torch.bfloat16)
        wp1, wp2 = ops.weight_pack(W1, bias=b1), ops.weight_pack(W2, transposed=True)
        drop = (0.15, 7)
        if rows == 0:
            t_a = graph_us(lambda: ops.gemm(L.GEMM_NT, x, W1, ab, N, f, d, ops.epilogue(bias=b1, act=L.ACT_SWISH, z=zb, drop=drop)))
            t_b = graph_us(lambda: ops.gemm(L.GEMM_NN, dy, W2, ab, N, f, d))
            tag = "tiled"
        else:
            t_a = graph_us(lambda: ops.gemm_panel(x, wp1, ab, N, f, d, ops.epilogue(act=L.ACT_SWISH, z=zb, drop=drop)))
            t_b = graph_us(lambda: ops.gemm_panel(dy, wp2, ab, N, f, d, ops.epilogue()))
            tag = f"p{rows}"
        print(f"d={d} N={N:6d} M={f:5d} {tag:6s} fwd+swish+Z+drop {t_a:7.1f}  dgrad-plain {t_b:7.1f}", flush=True)
