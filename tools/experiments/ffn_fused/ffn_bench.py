#!/usr/bin/env python3
"""Fused feed-forward module (smx_ffn_fwd / smx_ffn_bwd) against the two-GEMM path and fp32 torch math; timings.

    python tools/ffn_bench.py check          # correctness at small / ragged sizes (forward, backward)
    python tools/ffn_bench.py time [N]       # isolated timings at N frames (default 64000), fused vs unfused
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd import _lib as L, ops, functional as Fn

dev = "cuda"


def mk(N, D, F, seed=0, res_dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    x = r(N, D).to(dev).bfloat16()
    W1 = r(F, D, sc=D ** -0.5).to(dev).bfloat16(); b1 = r(F, sc=0.1).to(dev)
    W2 = r(D, F, sc=F ** -0.5).to(dev).bfloat16(); b2 = r(D, sc=0.1).to(dev)
    res = r(N, D).to(dev).to(res_dtype)
    gam = (1 + r(D, sc=0.1)).to(dev); bet = r(D, sc=0.1).to(dev)
    return x, W1, b1, W2, b2, res, gam, bet


def unfused(x, W1, b1, W2, b2, res, gam, bet, act, alpha, d1, d2):
    a, z = Fn.linear_fwd(x, W1, b1, act, None, save_z=True, drop=d1)
    post = []
    y, _ = Fn.linear_fwd(a, W2, b2, L.ACT_NONE, None, res=res, alpha=alpha, drop=d2, ln_next=(gam, bet, 1e-5, L.ACT_NONE, True), ln_post=post)
    return y, z, a, post[0]


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def check():
    ok = True
    for (N, F, act, p) in ((128, 1024, L.ACT_SWISH, 0.0), (300, 1024, L.ACT_SWISH, 0.0), (1000, 256, L.ACT_GELU, 0.0), (517, 1024, L.ACT_RELU, 0.0),
                           (4096, 1024, L.ACT_SWISH, 0.15), (33000, 1024, L.ACT_SWISH, 0.15)):
        D = 256
        x, W1, b1, W2, b2, res, gam, bet = mk(N, D, F, seed=N)
        d1 = (p, 1234567) if p > 0 else None
        d2 = (p, 7654321) if p > 0 else None
        y, z, a, (hy, st) = ops.ffn_fwd(x, W1, b1, W2, b2, act, res, 0.5, d1, d2, save_z=True, save_a=True, ln_next=(gam, bet, 1e-5, True))
        torch.cuda.synchronize()
        yu, zu, au, (hyu, stu) = unfused(x, W1, b1, W2, b2, res, gam, bet, act, 0.5, d1, d2)
        torch.cuda.synchronize()
        e = {"y": rel(y, yu), "z": rel(z, zu), "a": rel(a, au), "ln": rel(hy, hyu), "stats": rel(st, stu)}
        line = f"N={N} F={F} act={act} p={p}: vs unfused " + " ".join(f"{k}={v:.2e}" for k, v in e.items())
        if p == 0.0:
            xf, W1f, W2f = x.float(), W1.float(), W2.float()
            zr = xf @ W1f.t() + b1
            ar = {L.ACT_SWISH: torch.nn.functional.silu, L.ACT_GELU: torch.nn.functional.gelu, L.ACT_RELU: torch.relu}[act](zr)
            yr = res.float() + 0.5 * (ar.bfloat16().float() @ W2f.t() + b2)
            lr = torch.nn.functional.layer_norm(yr, (D,), gam, bet, 1e-5)
            e2 = {"y": rel(y, yr), "z": rel(z, zr), "ln": rel(hy, lr)}
            line += " | vs fp32 torch " + " ".join(f"{k}={v:.2e}" for k, v in e2.items())
            e.update({"r" + k: v for k, v in e2.items()})
        bad = any(not (v < 2e-2) for v in e.values())
        ok = ok and not bad
        print(("FAIL " if bad else "ok   ") + line, flush=True)
    print("ALL OK" if ok else "FAILED")
    return ok


def timing(N):
    D, F = 256, 1024
    x, W1, b1, W2, b2, res, gam, bet = mk(N, D, F)
    pad = int(os.environ.get("WPAD", "0"))
    if pad:                                   # experiment: weight rows with a non-power-of-two pitch (L2 channel spread)
        W1p = torch.zeros(F, D + pad, device=dev, dtype=torch.bfloat16); W1p[:, :D] = W1; W1 = W1p[:, :D]
        W2p = torch.zeros(D, F + pad, device=dev, dtype=torch.bfloat16); W2p[:, :F] = W2; W2 = W2p[:, :F]
    for p in (0.15, 0.0):
        d1 = (p, 1234567) if p > 0 else None
        d2 = (p, 7654321) if p > 0 else None
        tf = time_kernel(lambda: ops.ffn_fwd(x, W1, b1, W2, b2, L.ACT_SWISH, res, 0.5, d1, d2, save_z=True, ln_next=(gam, bet, 1e-5, True)), iters=20, warm=3)
        tu = time_kernel(lambda: unfused(x, W1, b1, W2, b2, res, gam, bet, L.ACT_SWISH, 0.5, d1, d2), iters=20, warm=3)
        nb = N * D * 2 * 4 + N * F * 2 + 2 * F * D * 2
        print(f"N={N} p={p}: fused fwd {tf*1e6:7.1f} us ({nb/tf/1e9:6.0f} GB/s algorithmic, {4.0*N*D*F/tf/1e12:5.0f} TF)   unfused pair {tu*1e6:7.1f} us", flush=True)
    tf = time_kernel(lambda: ops.ffn_fwd(x, W1, b1, W2, b2, L.ACT_SWISH, res, 0.5, None, None, save_z=False, ln_next=None), iters=20, warm=3)
    print(f"N={N} inference (no Z, no LN): fused fwd {tf*1e6:7.1f} us")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(0 if check() else 1)
    timing(int(sys.argv[2]) if len(sys.argv) > 2 else 64000)
