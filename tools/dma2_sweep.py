#!/usr/bin/env python3
"""NT / NN GEMM shapes of the C2b / C2a steps through the tiled kernel vs the all-DMA kernel (SMX_GEMM_DMA2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
print("DMA2", os.environ.get("SMX_GEMM_DMA2", "0"), flush=True)
N = 64000
for (K, M) in ((256, 1024), (1024, 256), (256, 256), (256, 512), (512, 256), (512, 2048), (2048, 512), (512, 512)):
    run(N, K, M, "NT", epi="swishz"); run(N, K, M, "NT", epi="bias"); run(N, K, M, "NN"); run(N, K, M, "NNag"); run(N, K, M, "NTres")
