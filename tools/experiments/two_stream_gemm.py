#!/usr/bin/env python3
"""Concept probe: do the phases of the fused GEMMs (MFMA-bound main loop, HBM-bound epilogue) overlap ACROSS kernels when two
half-batch chains run on two streams?  One FFN-like chain (up-projection +bias+Swish+Z+dropout, down-projection +bias+dropout
+fp32 residual+LayerNorm) on N frames in one stream vs. the same chain on two N/2 halves in two streams (offset by one launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summarymixing_amd import _lib as L, ops

def make(N, d=256, f=1024):
    bf = torch.bfloat16
    x = torch.randn(N, d, device="cuda").to(bf)
    w1 = (torch.randn(f, d, device="cuda") * 0.05).to(bf); b1 = torch.randn(f, device="cuda")
    w2 = (torch.randn(d, f, device="cuda") * 0.05).to(bf); b2 = torch.randn(d, device="cuda")
    y = torch.empty(N, f, device="cuda", dtype=bf); z = torch.empty(N, f, device="cuda", dtype=bf)
    r = torch.randn(N, d, device="cuda"); o = torch.empty(N, d, device="cuda")
    g_, b_ = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    hy = torch.empty(N, d, device="cuda", dtype=bf); st = torch.empty(N, 2, device="cuda")
    e1 = ops.epilogue(bias=b1, act=L.ACT_SWISH, z=z, drop=(0.15, 7))
    e2 = ops.epilogue(bias=b2, res=r, alpha=0.5, drop=(0.15, 99), out_mode=L.OUT_F32, ln_fwd=(g_, b_, hy, st, 1e-5, L.ACT_NONE))
    keep = (x, w1, b1, w2, b2, y, z, r, o, g_, b_, hy, st, e1, e2)
    def chain(reps):
        for _ in range(reps):
            ops.gemm(L.GEMM_NT, x, w1, y, N, f, d, e1)
            ops.gemm(L.GEMM_NT, y, w2, o, N, d, f, e2)
    return chain, keep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
REPS = 12
full, k0 = make(N)
ha, k1 = make(N // 2)
hb, k2 = make(N // 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def one_stream():
    full(REPS)
def two_streams():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ha(REPS)
    with torch.cuda.stream(sb):
        ops.gemm(L.GEMM_NT, k2[0], k2[1], k2[5], N // 2, 1024, 256, k2[13])    # one extra launch: the offset
        hb(REPS)
    cur.wait_stream(sa); cur.wait_stream(sb)
def halves_one_stream():
    ha(REPS); hb(REPS)
for name, fn in (("one stream, N frames", one_stream), ("two half chains, ONE stream", halves_one_stream), ("two half chains, TWO streams", two_streams)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); e1.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1) / 5 / REPS * 1e3:8.1f} us per FFN pair", flush=True)
