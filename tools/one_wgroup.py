"""Run the grouped wgrad of one C2b layer (8 weights, 64000 frames) a few times: profiling / PMC target."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import functional as F  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
which = sys.argv[2] if len(sys.argv) > 2 else "layer"
shapes = {"layer": [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)],
          "one": [(1024, 256)],
          "c2a": [(2048, 512), (512, 2048), (2048, 512), (512, 2048), (1024, 512), (512, 1024), (1024, 512), (512, 512)]}[which]
ops_ = [((torch.randn(rows, M, device="cuda") * 0.5).bfloat16(), torch.randn(rows, K, device="cuda").bfloat16(),
         torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")) for M, K in shapes]
import time  # noqa: E402
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0     # > 0: loop for that long (tools/power_trace.sh) and print the average
n, t0 = 0, time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while n < 8 or time.time() - t0 < secs:
    for _ in range(8):
        for dz, x, gW, gb in ops_:
            F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
        F.flush_deferred()
    n += 8
    torch.cuda.synchronize()
e1.record()
e1.synchronize()
if secs > 0:
    print(f"wgrad_group {which} {rows} frames (+ reduce_jobs) {e0.elapsed_time(e1) * 1e3 / n:8.1f} us per launch over {n} launches")
