#!/usr/bin/env python3
"""Yardstick only (never a dependency): hipBLASLt (through torch) PLUS the unfused pass over its output that this repo's fused
epilogues replace, at the four GEMM shapes that lead the C2b step (64 000 frames, bf16 operands, float32 residual stream), next
to the fused smx_gemm launch doing the same work.  The unfused passes are this repo's own standalone kernels (the best
single-pass implementation available here), so the comparison isolates FUSION, not kernel quality.

    python tools/blaslt_plus_epilogue.py > profiles/r03_blaslt_plus_epilogue.txt
    D=512 F=2048 python tools/blaslt_plus_epilogue.py > profiles/r05_blaslt_plus_epilogue_d512.txt   (the recipe width; round 4: the
    LayerNorm was a separate launch there - SMX_LN_FUSE=0 reproduces that: "fused" = the fused GEMM launch + the standalone
    LayerNorm kernel; round 5: the row-complete 128 x 512 tile)"""
import os
import sys

import torch
import torch.nn.functional as tF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel                                  # noqa: E402
from summarymixing_amd import _lib as L, ops                    # noqa: E402

N, d, f = 64000, int(os.environ.get("D", 256)), int(os.environ.get("F", 1024))
LNF = L.lib().smx_gemm_ln_fused_ok(L.BF16, N, d, f) == 1 and os.environ.get("SMX_LN_FUSE", "1") != "0"   # LayerNorm inside the GEMM epilogue (row-complete 128 x 256 / 128 x 512 tile)
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x, h = rnd(N, d).bfloat16(), rnd(N, f).bfloat16()
W1, W2 = (rnd(f, d) * 0.06).bfloat16(), (rnd(d, f) * 0.03).bfloat16()
b1, b2 = rnd(f) * 0.1, rnd(d) * 0.1
res32 = rnd(N, d)
gam, bet = rnd(d) * 0.3 + 1, rnd(d) * 0.3
dy = rnd(N, d).bfloat16()
z = rnd(N, f).bfloat16()
def T(fn):
    """us per call: the median of three runs of 30 back-to-back launches, each behind 8 warm-up launches (round 5: one run of 20
    behind 4 warm-ups read 5-15 % high for whatever was measured first after a pause - clocks and caches cold)."""
    return sorted(time_kernel(fn, iters=30, warm=8) * 1e6 for _ in range(3))[1]


_wa, _wb = torch.randn(8192, 8192, device=dev).bfloat16(), torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(60):                                            # ~0.1 s of matrix work: the chip at its loaded clocks before the first row
    torch.matmul(_wa, _wb)
del _wa, _wb
print(f"# hipBLASLt + unfused pass vs the fused launch of this library (rows 1-2: smx_gemm_panel from round 5 on; rows 3-4: smx_gemm), N = {N} frames, d = {d}, d_ffn = {f}, bf16, dropout 0.15 (us)")
print(f"# {'kernel':66s} {'hipBLASLt':>9s} {'+ pass':>8s} {'= sum':>8s} {'fused':>8s} {'fused/sum':>9s}")

# 1) FFN up-projection: z = x W1^T + b1 (saved), a = D(Swish(z))
lin = T(lambda: tF.linear(x, W1, b1.bfloat16()))
zz = tF.linear(x, W1, b1.bfloat16())
a_out = torch.empty_like(zz)


def up_pass():
    a = tF.silu(zz)
    ops.dropout(a, 0.15, 7, out=a_out)


p1 = T(up_pass)
zb, ab = torch.empty(N, f, device=dev, dtype=torch.bfloat16), torch.empty(N, f, device=dev, dtype=torch.bfloat16)
PANEL = ops.gemm_panel_ok(x, f, d) and os.environ.get("SMX_PANEL", "1") != "0"     # what functional.linear_fwd / linear_bwd launch for these two rows
if PANEL:
    wp1, wp2t = ops.weight_pack(W1, bias=b1), ops.weight_pack(W2, transposed=True)
    fu = T(lambda: ops.gemm_panel(x, wp1, ab, N, f, d, ops.epilogue(act=L.ACT_SWISH, z=zb, drop=(0.15, 7))))
else:
    fu = T(lambda: ops.gemm(L.GEMM_NT, x, W1, ab, N, f, d, ops.epilogue(bias=b1, act=L.ACT_SWISH, z=zb, drop=(0.15, 7))))
print(f"  {f'NT {d}->{f} + bias + Swish + Z + dropout':66s} {lin:9.1f} {p1:8.1f} {lin + p1:8.1f} {fu:8.1f} {fu / (lin + p1):9.2f}")

# 2) act-grad dgrad: dz = D(g * Swish'(z)), g = dy W2
mm = T(lambda: torch.matmul(dy, W2))
g = torch.matmul(dy, W2)
dz = torch.empty_like(g)
p2 = T(lambda: ops.act_mask_bwd(g, z, None, L.ACT_SWISH, 1.0, dz, None, None, 0, (0.15, 7)))
dzb = torch.empty(N, f, device=dev, dtype=torch.bfloat16)
if PANEL:
    fu = T(lambda: ops.gemm_panel(dy, wp2t, dzb, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 7))))
else:
    fu = T(lambda: ops.gemm(L.GEMM_NN, dy, W2, dzb, N, f, d, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 7))))
print(f"  {f'NN {d}->{f} + act-grad(z) + dropout':66s} {mm:9.1f} {p2:8.1f} {mm + p2:8.1f} {fu:8.1f} {fu / (mm + p2):9.2f}")

# 3) FFN down-projection + residual (float32 stream) + dropout + LayerNorm of the new stream tensor
lin = T(lambda: tF.linear(h, W2, b2.bfloat16()))
o = tF.linear(h, W2, b2.bfloat16())
od = torch.empty_like(o)
stream = torch.empty(N, d, device=dev)


def down_pass():
    ops.dropout(o, 0.15, 9, out=od)
    torch.add(res32, od, alpha=0.5, out=stream)
    ops.layernorm_fwd(stream, gam, bet, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16)


p3 = T(down_pass)
outs, hy, st = torch.empty(N, d, device=dev), torch.empty(N, d, device=dev, dtype=torch.bfloat16), torch.empty(N, 2, device=dev)
if LNF:
    fu = T(lambda: ops.gemm(L.GEMM_NT, h, W2, outs, N, d, f, ops.epilogue(bias=b2, res=res32, alpha=0.5, drop=(0.15, 9), out_mode=L.OUT_F32,
                                                                        ln_fwd=(gam, bet, hy, st, 1e-5, L.ACT_NONE))))
else:
    def down_fused():
        ops.gemm(L.GEMM_NT, h, W2, outs, N, d, f, ops.epilogue(bias=b2, res=res32, alpha=0.5, drop=(0.15, 9), out_mode=L.OUT_F32))
        ops.layernorm_fwd(outs, gam, bet, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16)
    fu = T(down_fused)
print(f"  {f'NT {f}->{d} + bias + dropout + fp32 residual + LayerNorm':66s} {lin:9.1f} {p3:8.1f} {lin + p3:8.1f} {fu:8.1f} {fu / (lin + p3):9.2f}")

# 4) dgrad of the up-projection + LayerNorm backward (+ residual gradient)
dzu = rnd(N, f).bfloat16()
mm = T(lambda: torch.matmul(dzu, W1))
gh = torch.matmul(dzu, W1)
xs = rnd(N, d)
_, stats = ops.layernorm_fwd(xs, gam, bet, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16)
dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
p4 = T(lambda: ops.layernorm_bwd(gh, xs, gam, bet, stats, dg, db, dy, L.ACT_NONE))
tr = L.lib().smx_gemm_ln_tile_rows_for(N, d)
ws = torch.zeros(((N + tr - 1) // tr) * 2 * d, device=dev)
dxo = torch.empty(N, d, device=dev, dtype=torch.bfloat16)
if LNF:
    fu = T(lambda: ops.gemm(L.GEMM_NN, dzu, W1, dxo, N, d, f, ops.epilogue(res=dy, ln_bwd=(xs, stats, gam, ws, None, None, None, True))))
else:
    ghb = torch.empty(N, d, device=dev, dtype=torch.bfloat16)

    def dgrad_fused():
        ops.gemm(L.GEMM_NN, dzu, W1, ghb, N, d, f)
        ops.layernorm_bwd(ghb, xs, gam, bet, stats, dg, db, dy, L.ACT_NONE)
    fu = T(dgrad_fused)
print(f"  {f'NN {f}->{d} + LayerNorm backward (fp32 rows) + residual gradient':66s} {mm:9.1f} {p4:8.1f} {mm + p4:8.1f} {fu:8.1f} {fu / (mm + p4):9.2f}")
