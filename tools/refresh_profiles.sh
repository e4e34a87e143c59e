#!/usr/bin/env bash
# Re-measure everything profiles/ holds (run on the GPU box through gpurun; outputs land in gpurun_out/refresh/,
# copy them into profiles/ afterwards):  gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh v7'
set -uo pipefail
TAG="${1:-vX}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/refresh"; mkdir -p "$OUT"
cd "$ROOT"
python bench.py 2>/dev/null | tail -1 > "$OUT/r01_bench_default.json"
python bench.py --batch 64 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/r01_bench_b64.json"
python bench.py --config c2a --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/r01_bench_c2a.json"
python bench.py --mode forward --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/r01_bench_fwd.json"
python bench.py --config c4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/r01_bench_c4.json"
python bench.py --config c5 --steps 5 --warmup 2 2>/dev/null | tail -1 > "$OUT/r01_bench_c5.json"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_refresh && rocprofv3 --kernel-trace --stats -d /tmp/prof_refresh -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
DB=$(find /tmp/prof_refresh -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline   (7 steps incl. warmup; C2b, B=128 x T=500, bf16, dropout 0.15)"; python tools/prof_summary.py "$DB" 7; echo; echo "## GEMM launches by grid (shape)"; python tools/prof_by_grid.py "$DB" 7; } > "$OUT/r01_step_c2b_${TAG}.txt"
bash tools/pmc_traffic.sh > "$OUT/r01_pmc_traffic.txt" 2>&1
ls -la "$OUT"
