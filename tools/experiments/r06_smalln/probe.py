#!/usr/bin/env python3
"""Round 6: what would split-K over WORKGROUPS buy the small-grid GEMMs?  Emulated with the product kernel: a BATCHED GEMM whose batch
index walks the K slices (A + b * K/S columns, W + b * K/S columns, fp32 slab b) is exactly the split-K main loop + slab store.
Prints isolated launch times at N frames (default 3750): the product dispatch, and S = 2 / 4 / 8 slices on the tile the dispatch picks."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops
from bench import time_kernel

N = int(os.environ.get("N", 3750))


def t_us(fn):
    return sorted(time_kernel(fn, iters=40, warm=10) for _ in range(3))[1] * 1e6


def probe(layout, K, M):
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16() if layout == "NT" else (torch.randn(K, M, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    lay = L.GEMM_NT if layout == "NT" else L.GEMM_NN
    base = t_us(lambda: ops.gemm(lay, x, w, y, N, M, K))
    sym = ops.gemm_symbol(lay, x, w, y, N, M, K)
    line = f"{layout} N={N} K={K} M={M}: product {base:6.1f} us [{sym.split('<')[1][:40]}]"
    for S in (2, 4, 8):
        if K // S < 128:
            continue
        ks = K // S
        slab = torch.empty(S, N, M, device="cuda", dtype=torch.float32)
        e = ops.epilogue(out_mode=L.OUT_F32)
        if layout == "NT":   # W (M, K): slice b = columns [b ks, (b+1) ks)
            fn = lambda: ops.gemm(lay, x[:, :ks], w[:, :ks], slab[0], N, M, ks, e, batch=S, sa=ks, sb=ks, sc=N * M, lda=K, ldb=K, ldc=M)
        else:                # W (K, M): slice b = rows [b ks, (b+1) ks)
            fn = lambda: ops.gemm(lay, x[:, :ks], w[:ks], slab[0], N, M, ks, e, batch=S, sa=ks, sb=ks * M, sc=N * M, lda=K, ldb=M, ldc=M)
        fn(); torch.cuda.synchronize()
        ref = (x.float() @ (w.float().t() if layout == "NT" else w.float()))
        err = float((slab.sum(0) - ref).abs().max() / ref.abs().max())
        s2 = ops.gemm_symbol(lay, x[:, :ks], w[:, :ks] if layout == "NT" else w[:ks], slab[0], N, M, ks, e, S, ks, ks if layout == "NT" else ks * M, N * M, 1, K, K if layout == "NT" else M, M)
        line += f" | S={S} {t_us(fn):6.1f} us ({s2.split('<')[1].split(',')[3].strip()}x{s2.split('<')[1].split(',')[4].strip()}, err {err:.0e})"
    print(line, flush=True)


if __name__ == "__main__":
    print(f"# lib {L.LIB_PATH}")
    d = int(os.environ.get("D", 512))
    for layout in ("NT", "NN"):
        probe(layout, 4 * d, d); probe(layout, 2 * d, d); probe(layout, d, d); probe(layout, d, 4 * d); probe(layout, d, 2 * d)
