#!/usr/bin/env python3
"""The weight gradients of one encoder layer: grouped split-K slabs + smx_reduce_jobs (smx_wgrad_group) against the slab-free form
(smx_wgrad_group_direct), inside a replayed hipGraph.   python tools/experiments/r06_smalln/wgrad_direct_bench.py [c2a|layer]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import functional as F
from tools.graph_timer import graph_us

which = sys.argv[1] if len(sys.argv) > 1 else "c2a"
shapes = {"layer": [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)],
          "c2a": [(2048, 512), (512, 2048), (2048, 512), (512, 2048), (1024, 512), (512, 1024), (1024, 512), (512, 512)]}[which]
for rows in [int(v) for v in os.environ.get("NS", "500,2000,3750,6000,8000,12000,15000,24000,32000").split(",")]:
    ops_ = [((torch.randn(rows, M, device="cuda") * 0.5).bfloat16(), torch.randn(rows, K, device="cuda").bfloat16(),
             torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")) for M, K in shapes]

    def run():
        for dz, x, gW, gb in ops_:
            F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
        F.flush_deferred()
    res = {}
    for name, mx in (("slabs+reduce", 0), ("direct", 1 << 30)):
        F._Deferred.group_direct_max_rows = mx
        for _, _, gW, gb in ops_:
            gW.zero_(); gb.zero_()
        run(); torch.cuda.synchronize()
        res[name] = ([o[2].clone() for o in ops_], [o[3].clone() for o in ops_], graph_us(run, reps=10))
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res["direct"][0] + res["direct"][1], res["slabs+reduce"][0] + res["slabs+reduce"][1]))
    print(f"{which} {rows:6d} frames: slabs + reduce_jobs {res['slabs+reduce'][2]:7.1f} us | direct {res['direct'][2]:7.1f} us   (max rel diff {err:.1e})", flush=True)
