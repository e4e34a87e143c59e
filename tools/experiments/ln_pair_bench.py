import sys, os, torch
sys.path.insert(0, "/root/repo")
from summarymixing_amd import ops
from bench import time_kernel
for N, D in ((64000, 512), (240000, 512), (64000, 256), (1, 64), (37, 144)):
    x = torch.randn(N, D, device="cuda") * 3 + 0.7
    g1, b1, g2, b2 = [torch.randn(D, device="cuda") * 0.3 + 1 for _ in range(4)]
    for st in (True, False):
        tp = time_kernel(lambda: ops.layernorm_fwd_pair(x, g1, b1, 1e-5, g2, b2, 1e-5, st, torch.bfloat16)) * 1e6
        def two():
            r1, _ = ops.layernorm_fwd(x, g1, b1, 1e-5, st)
            ops.layernorm_fwd(r1, g2, b2, 1e-5, st, out_dtype=torch.bfloat16)
        t2 = time_kernel(two) * 1e6
        print(f"N={N} D={D} stats={st}: pair {tp:.1f} us, two launches {t2:.1f} us")
    y1, s1, y2, s2 = ops.layernorm_fwd_pair(x, g1, b1, 1e-5, g2, b2, 1e-6, True, torch.bfloat16)
    r1, t1 = ops.layernorm_fwd(x, g1, b1, 1e-5, True)
    r2, t2_ = ops.layernorm_fwd(r1, g2, b2, 1e-6, True, out_dtype=torch.bfloat16)
    print("   y1 eq", torch.equal(y1, r1), float((y1 - r1).abs().max()), "s1 eq", torch.equal(s1, t1), float((s1 - t1).abs().max()),
          "y2 eq", torch.equal(y2, r2), float((y2.float() - r2.float()).abs().max()), "s2 eq", torch.equal(s2, t2_), float((s2 - t2_).abs().max()))
