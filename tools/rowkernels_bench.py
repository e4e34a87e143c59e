#!/usr/bin/env python3
"""Micro-benchmark of the row kernels (everything that is not a GEMM) at the shapes of the C2b / C2a training step:
B = 128 utterances x T = 500 frames, HIP-event timing, algorithmic bytes per launch and the rate they imply."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops  # noqa: E402
from bench import time_kernel  # noqa: E402

B, T = 128, 500
N = B * T
dev = "cuda"
bf = torch.bfloat16


def line(name, t, nbytes):
    print(f"{name:64s} {t*1e6:8.1f} us  {nbytes/1e6:7.0f} MB  {nbytes/t/1e12:5.2f} TB/s", flush=True)


def run(d):
    print(f"# d_model = {d}")
    k = 31
    mask = (torch.rand(N, device=dev) < 0.75).view(torch.uint8)
    # the cell's summary: pool, broadcast (+ dropout) into the merge input, broadcast of the gradient (+ act / mask backward)
    g2 = torch.randn(N, 2 * d, device=dev).to(bf)
    s = g2[:, d:]
    line("masked_mean pool (s columns of g, ld 2d)", time_kernel(lambda: ops.masked_mean(s, mask, B, T, scale=True, want_inv=True)), N * d * 2 + N)
    sbar, inv = ops.masked_mean(s, mask, B, T, scale=True, want_inv=True)
    cat = torch.empty(N, 2 * d, device=dev, dtype=bf)
    line("bcast_rows + dropout -> cat[:, d:] (ld 2d)", time_kernel(lambda: ops.bcast_rows(sbar, None, cat[:, d:], B, T, drop=(0.15, 77))), N * d * 2)
    line("bcast_rows plain -> cat[:, d:]", time_kernel(lambda: ops.bcast_rows(sbar, None, cat[:, d:], B, T)), N * d * 2)
    dg = torch.empty(N, 2 * d, device=dev, dtype=bf)
    line("bcast_rows + act'(z) * mask -> dg[:, d:] (z ld 2d)", time_kernel(lambda: ops.bcast_rows_act_bwd(sbar, inv, dg[:, d:], B, T, s, mask, L.ACT_SWISH)), 2 * N * d * 2 + N)
    # conv module: GLU + depthwise conv k = 31, forward / backward
    p = torch.randn(N, 2 * d, device=dev).to(bf)
    wd = torch.randn(d, k, device=dev) * 0.1
    bd = torch.randn(d, device=dev)
    line("dwconv GLU fwd k=31", time_kernel(lambda: ops.dwconv_fwd(p, wd, bd, B, T, d, k, True, L.PAD_ZERO, 0)), N * d * 2 * 3)
    dy = torch.randn(N, d, device=dev).to(bf)
    ws = torch.empty(L.lib().smx_dwconv1d_glu_bwd_workspace(B, T, d, k), dtype=torch.uint8, device=dev)
    line("dwconv GLU bwd k=31 (deferred tap partials)", time_kernel(lambda: ops.dwconv_bwd(dy, p, wd, bd, None, None, B, T, d, k, True, L.PAD_ZERO, 0, ws=ws)), N * d * 2 * 5)
    # LayerNorms that stay standalone: fp32 stream -> bf16, bf16 -> bf16 (+ Swish), backward (+ residual gradient)
    x32 = torch.randn(N, d, device=dev)
    xb = x32.to(bf)
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    line("layernorm fwd fp32 -> bf16", time_kernel(lambda: ops.layernorm_fwd(x32, gam, bet, 1e-5, True, out_dtype=bf)), N * d * 6)
    line("layernorm fwd bf16 -> bf16 + Swish", time_kernel(lambda: ops.layernorm_fwd(xb, gam, bet, 1e-5, True, act=L.ACT_SWISH)), N * d * 4)
    _, st = ops.layernorm_fwd(x32, gam, bet, 1e-5, True, out_dtype=bf)
    lws = torch.empty(L.lib().smx_layernorm_bwd_workspace(N, d), dtype=torch.uint8, device=dev)
    r = torch.randn(N, d, device=dev).to(bf)
    line("layernorm bwd (x fp32) + res", time_kernel(lambda: ops.layernorm_bwd(dy, x32, gam, bet, st, None, None, res=r, ws=lws)), N * d * (2 + 4 + 2 + 2))
    # the same on four rotating operand sets (> 1 GB: nothing of the previous launch is left in the 256 MB MALL - the in-step case)
    sets = [(torch.randn(N, d, device=dev).to(bf), torch.randn(N, d, device=dev), torch.randn(N, d, device=dev).to(bf),
             torch.empty(N, d, device=dev, dtype=bf)) for _ in range(4)]
    cnt = [0]

    def rot():
        a, b_, c, o = sets[cnt[0] & 3]
        cnt[0] += 1
        ops.layernorm_bwd(a, b_, gam, bet, st, None, None, res=c, ws=lws, dx_out=o)
    line("layernorm bwd (x fp32) + res, cold operands", time_kernel(rot, iters=40), N * d * (2 + 4 + 2 + 2))
    line("layernorm bwd (x bf16) + Swish", time_kernel(lambda: ops.layernorm_bwd(dy, xb, gam, bet, st, None, None, act=L.ACT_SWISH, ws=lws)), N * d * (2 + 2 + 2))


def run_csgu():
    """the Branchformer's CSGU LayerNorm over 1536 channels (C4: 128 x 250 frames), bf16 in and out"""
    n, d = 32000, 1536
    print(f"# CSGU LayerNorm, {n} x {d}")
    sets = [(torch.randn(n, d, device=dev).to(bf), torch.randn(n, d, device=dev).to(bf)) for _ in range(4)]
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    cnt = [0]

    def fwd():
        cnt[0] += 1
        ops.layernorm_fwd(sets[cnt[0] & 3][0], gam, bet, 1e-5, True)
    line("layernorm fwd bf16 (32000 x 1536), cold operands", time_kernel(fwd, iters=40), n * d * 4)
    _, st = ops.layernorm_fwd(sets[0][0], gam, bet, 1e-5, True)
    lws = torch.empty(L.lib().smx_layernorm_bwd_workspace(n, d), dtype=torch.uint8, device=dev)
    out = torch.empty(n, d, device=dev, dtype=bf)

    def bwd():
        cnt[0] += 1
        x, dy = sets[cnt[0] & 3]
        ops.layernorm_bwd(dy, x, gam, bet, st, None, None, ws=lws, dx_out=out)
    line("layernorm bwd bf16 (32000 x 1536), cold operands", time_kernel(bwd, iters=40), n * d * 6)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "csgu":
        run_csgu()
        sys.exit(0)
    for d in (256, 512):
        run(d)
    run_csgu()
