#!/usr/bin/env python3
"""wgrad (TN) GEMM shapes of the C2b step with SMX_GEMM_ABLATE (1 no stores, 2 no MFMA, 4 no loads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
print("ABLATE", os.environ.get("SMX_GEMM_ABLATE", "0"))
run(64000, 1024, 256, "TN"); run(64000, 256, 1024, "TN"); run(64000, 256, 256, "TN"); run(64000, 512, 256, "TN")
