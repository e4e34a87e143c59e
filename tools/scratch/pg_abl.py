import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.gemm_bench import run
print("NOROT(ablate bits)", os.environ.get("SMX_GEMM_NOROT"))
run(64000, 1024, 256, "NN"); run(64000, 1024, 256, "NT", epi="plain"); run(64000, 2048, 512, "NN")
