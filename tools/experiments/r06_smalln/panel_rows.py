#!/usr/bin/env python3
"""Panel GEMM with 128 / 64 / 32-row panels (SMX_PANEL_ROWS, read once per process) against the tiled smx_gemm at small N, timed inside
a replayed hipGraph (tools/graph_timer.py).   N=3750 D=512 python tools/experiments/r06_smalln/panel_rows.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops
from tools.graph_timer import graph_us

N, d = int(os.environ.get("N", 3750)), int(os.environ.get("D", 512))
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x, dy = rnd(N, d).bfloat16(), rnd(N, d).bfloat16()
mk = (torch.rand(N, device=dev) < 0.8).view(torch.uint8)
rows = L.get_config()["panel_rows"]
print(f"# N = {N}, K = d = {d}, SMX_PANEL_ROWS = {rows or 'auto'}: us per launch inside a replayed graph (tiled | panel)")
for f in (4 * d, 2 * d, d):
    W1, W2 = (rnd(f, d) * 0.06).bfloat16(), (rnd(d, f) * 0.03).bfloat16()
    b1 = rnd(f) * 0.1
    z = rnd(N, f).bfloat16()
    zb, ab, ab2 = torch.empty(N, f, device=dev, dtype=torch.bfloat16), torch.empty(N, f, device=dev, dtype=torch.bfloat16), torch.empty(N, f, device=dev, dtype=torch.bfloat16)
    wp1, wp2 = ops.weight_pack(W1, bias=b1), ops.weight_pack(W2, transposed=True)
    drop = (0.15, 7)
    cases = [
        ("NT +bias+swish+Z+drop", lambda o: ops.gemm(L.GEMM_NT, x, W1, o, N, f, d, ops.epilogue(bias=b1, act=L.ACT_SWISH, z=zb, drop=drop)),
         lambda o: ops.gemm_panel(x, wp1, o, N, f, d, ops.epilogue(act=L.ACT_SWISH, z=zb, drop=drop))),
        ("NT +bias+swish+Z+mask", lambda o: ops.gemm(L.GEMM_NT, x, W1, o, N, f, d, ops.epilogue(bias=b1, act=L.ACT_SWISH, z=zb, row_mask=mk)),
         lambda o: ops.gemm_panel(x, wp1, o, N, f, d, ops.epilogue(act=L.ACT_SWISH, z=zb, row_mask=mk))),
        ("NT +bias", lambda o: ops.gemm(L.GEMM_NT, x, W1, o, N, f, d, ops.epilogue(bias=b1)),
         lambda o: ops.gemm_panel(x, wp1, o, N, f, d, ops.epilogue())),
        ("NN actgrad+drop", lambda o: ops.gemm(L.GEMM_NN, dy, W2, o, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=drop)),
         lambda o: ops.gemm_panel(dy, wp2, o, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=drop))),
        ("NN plain", lambda o: ops.gemm(L.GEMM_NN, dy, W2, o, N, f, d),
         lambda o: ops.gemm_panel(dy, wp2, o, N, f, d, ops.epilogue())),
    ]
    for name, tiled, panel in cases:
        tiled(ab); panel(ab2); torch.cuda.synchronize()
        err = float((ab.float() - ab2.float()).abs().max() / ab.float().abs().max())
        print(f"  M={f:5d} {name:24s} {graph_us(lambda: tiled(ab)):6.1f} | {graph_us(lambda: panel(ab2)):6.1f}   (max diff {err:.1e})", flush=True)
