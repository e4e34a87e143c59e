"""Grouped wgrad (smx_wgrad_group) vs one slab GEMM per weight, isolated, at the bench's frame count.
    python tools/wgroup_bench.py [rows]         (SMX_WGROUP_BK=32|64, SMX_WGROUP_BLOCKS=N to vary the kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import functional as F, ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
SETS = {
    "c2b layer (8 weights)": [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)],
    "ffn pair": [(1024, 256), (256, 1024)],
    "one 1024x256": [(1024, 256)],
    "c2a layer (8 weights)": [(2048, 512), (512, 2048), (2048, 512), (512, 2048), (1024, 512), (512, 1024), (1024, 512), (512, 512)],
}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, shapes in SETS.items():
    if "c2a" in name and rows > 64000:
        continue
    ops_ = [((torch.randn(rows, M, device="cuda") * 0.5).bfloat16(), torch.randn(rows, K, device="cuda").bfloat16(),
             torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")) for M, K in shapes]
    nb = sum((M + K) * rows * 2 + 4 * M * K for M, K in shapes)
    fl = sum(2.0 * rows * M * K for M, K in shapes)

    def grouped():
        for dz, x, gW, gb in ops_:
            F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
        F.flush_deferred()

    def grouped_nored():
        for dz, x, gW, gb in ops_:
            F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
        F._launch_groups()
        F._Deferred.jobs = []
        F._Deferred.pending.clear()

    def single():
        for dz, x, gW, gb in ops_:
            ops.wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], dbias=gb)
    tg, tn, ts = timeit(grouped), timeit(grouped_nored), timeit(single)
    print(f"{name:24s} rows={rows}: grouped {tg:8.1f} us ({nb / tg / 1e6:6.2f} TB/s alg, {fl / tg / 1e6:6.1f} TFLOP/s)  "
          f"kernel only {tn:8.1f} us   per-weight slabs+reduce {ts:8.1f} us", flush=True)
