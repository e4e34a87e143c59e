#!/usr/bin/env bash
# kernel-trace summary of the C2b / C2a step (panel GEMM on): gpurun -- 'bash tools/experiments/prof_c2b.sh [c2a]'
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; mkdir -p gpurun_out
CFG="${1:-}"; ARGS=""; [[ -n "$CFG" ]] && ARGS="--config $CFG"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_x && rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o p -- python "$ROOT/bench.py" $ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points > /dev/null 2>&1)
DB=$(find /tmp/prof_x -name '*.db' | head -1)
python tools/prof_summary.py "$DB" 7 | head -40
