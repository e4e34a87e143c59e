import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
run(32000, 3072, 512, "TN"); run(32000, 512, 3072, "TN"); run(32000, 512, 512, "TN"); run(32000, 1536, 512, "TN")
