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
for _ in range(8):
    for dz, x, gW, gb in ops_:
        F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
    F.flush_deferred()
torch.cuda.synchronize()
