#!/usr/bin/env bash
# HBM traffic (TCC FETCH_SIZE x2-corrected + WRITE_SIZE, separate --pmc passes) of the kernels round 6 added or re-tiled:
# the 64 x 256 row-complete LayerNorm tile at B = 64 and the slab-free grouped wgrad (whose FETCH_SIZE counts every XCD's own copy of the operands).
#   gpurun --timeout 1200 -- 'bash tools/pmc_r06.sh > gpurun_out/r06_pmc_traffic.txt 2>&1'
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
eval "$(sed -n '/^run() {/,/^}/p' "$ROOT/tools/pmc_traffic.sh" | sed 's/rocprofv3 --pmc/timeout 300 rocprofv3 --pmc/')"
cd /tmp; export TMPDIR=/tmp
run "@ [FFN down-projection -> norm1, 32 000 rows: the 64 x 256 tile]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTln 32000 1024 256
run "@ [dgrad K=1024 + LayerNorm backward, 32 000 rows: the 64 x 256 tile]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NNlnb 32000 1024 256
run "@ [conv-module out-projection, 32 000 rows: the 64 x 256 tile]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTlnc 32000 256 256
run "wgrad_group_direct bf16, C2a layer at 3750 frames [algorithmic: operands 2 x 3750 x 15872 B = 119 MB + dW 8 x RMW]" wgrad_group_direct_kernel python3 "$ROOT/tools/one_wgroup.py" 3750 c2a
