#!/usr/bin/env python
"""Determinism / correctness stress of smx_gemm_panel at full-chip sizes: every configuration is run 12 times on the same inputs,
every output must be bit-identical to the first and close to the tiled kernel's."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summarymixing_amd import _lib as L, ops
torch.manual_seed(0)
bad = 0
for N in (33000, 64000):
    for K, M in ((256, 512), (256, 1024), (512, 2048), (512, 1536)):
        x = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
        W = ((torch.rand(M, K, device="cuda") * 2 - 1) * 0.08).bfloat16()
        b = torch.rand(M, device="cuda") - 0.5
        z = (torch.rand(N, M, device="cuda") * 6 - 3).bfloat16()
        Wt = W.t().contiguous()
        wp, wpt, wp0 = ops.weight_pack(W, bias=b), ops.weight_pack(Wt, transposed=True), ops.weight_pack(W)
        cases = {
            "bias": (lambda o, zz: ops.gemm_panel(x, wp, o, N, M, K, ops.epilogue()), lambda o, zz: ops.gemm(L.GEMM_NT, x, W, o, N, M, K, ops.epilogue(bias=b)), False),
            "swishZ+drop": (lambda o, zz: ops.gemm_panel(x, wp, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, z=zz, drop=(0.15, 5))),
                            lambda o, zz: ops.gemm(L.GEMM_NT, x, W, o, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=zz, drop=(0.15, 5))), True),
            "actgrad+drop": (lambda o, zz: ops.gemm_panel(x, wpt, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 5))),
                             lambda o, zz: ops.gemm(L.GEMM_NN, x, Wt, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 5))), False),
            "plain dgrad": (lambda o, zz: ops.gemm_panel(x, wpt, o, N, M, K, ops.epilogue()), lambda o, zz: ops.gemm(L.GEMM_NN, x, Wt, o, N, M, K, ops.epilogue()), False),
        }
        for name, (fp, ft, hasz) in cases.items():
            o0, z0 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16), torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
            ot, zt = torch.empty_like(o0), torch.empty_like(o0)
            fp(o0, z0); ft(ot, zt)
            ref = float((o0.float() - ot.float()).abs().max() / ot.float().abs().max())
            nd = 0
            for it in range(12):
                o1, z1 = torch.full_like(o0, 3.0), torch.full_like(o0, 3.0)
                fp(o1, z1)
                d = int((o1 != o0).sum()) + (int((z1 != z0).sum()) if hasz else 0)
                nd += d > 0
                if d:
                    w = (o1 != o0).nonzero()
                    print(f"   run {it}: {d} differing elements, first at {w[0].tolist() if len(w) else '-'} rows {sorted(set((w[:, 0] // 128).tolist()))[:8]} cols {sorted(set((w[:, 1] // 64).tolist()))[:8]}")
            bad += nd > 0 or ref > 2e-2
            print(f"N={N} K={K} M={M} {name:14s}: vs tiled max-rel {ref:.2e}; nondeterministic runs {nd}/12")
print("FAILED" if bad else "ok")
