"""A/B of the persistent LDS-DMA GEMM (pgemm.hip) against the tiled kernels on the step's hot shapes.
Run twice: SMX_PGEMM=0 python tools/pgemm_ab.py ; SMX_PGEMM=1 python tools/pgemm_ab.py   (the knob is read once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run  # noqa: E402

N = int(os.environ.get("N", 64000))
print("SMX_PGEMM =", os.environ.get("SMX_PGEMM", "1"), flush=True)
for d in (256, 512):
    f = 4 * d
    run(N, d, f, "NT")                     # FFN up-projection +bias+swish+Z
    run(N, d, f, "NT", epi="plain")        # bias only
    run(N, d, 2 * d, "NT")                 # global_proj
    run(N, f, d, "NTres")                  # FFN down-projection + residual + dropout
    run(N, 2 * d, d, "NTres")              # merge (cat, K = 2d)
    run(N, d, d, "NTres")                  # conv out-projection
    run(N, d, f, "NNag")                   # fused act-grad dgrad
    run(N, f, d, "NN")                     # dh = dz1 W1
    run(N, 2 * d, d, "NN")                 # K = 2d dgrads
    run(N, d, d, "NN")
    run(N, d, 2 * d, "NN")
