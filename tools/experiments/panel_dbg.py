import os, sys, torch
sys.path.insert(0, os.getcwd())
from summarymixing_amd import _lib as L, ops
torch.manual_seed(0)
for N in (2500, 16000, 33000):
    K, M = 256, 512
    x = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    W = ((torch.rand(M, K, device="cuda") * 2 - 1) * 0.08).bfloat16()
    b = torch.rand(M, device="cuda") - 0.5
    wp = ops.weight_pack(W, bias=b)
    o = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); ot = torch.empty_like(o)
    ops.gemm_panel(x, wp, o, N, M, K, ops.epilogue()); ops.gemm(L.GEMM_NT, x, W, ot, N, M, K, ops.epilogue(bias=b))
    ref = (x.double() @ W.double().t() + b.double())
    e1 = (o.double() - ref).abs(); e2 = (ot.double() - ref).abs()
    print(N, "panel max err", float(e1.max()), "tiled max err", float(e2.max()), "ref max", float(ref.abs().max()))
    bad = (e1 > 0.05).nonzero()
    print("  bad elements", len(bad), "rows%128", sorted(set((bad[:, 0] % 128).tolist()))[:20], "panels", sorted(set((bad[:, 0] // 128).tolist()))[:10], "cols", sorted(set((bad[:, 1]).tolist()))[:20])
