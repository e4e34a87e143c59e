#!/usr/bin/env bash
# One rocprofv3 --kernel-trace --stats summary: bash tools/prof_one.sh <out file> <steps> "<header>" <command...>
# (run on the GPU box through gpurun; the summary is tools/prof_summary.py + tools/prof_by_grid.py of the rocpd database)
set -uo pipefail
OUTF="$1"; STEPS="$2"; HDR="$3"; shift 3
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$(dirname "$OUTF")"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_one && rocprofv3 --kernel-trace --stats -d /tmp/prof_one -o p -- "$@" > /tmp/prof_one.log 2>&1)
DB=$(find /tmp/prof_one -name '*.db' | head -1)
{ echo "# $HDR"; python "$ROOT/tools/prof_summary.py" "$DB" "$STEPS"; echo; echo "## GEMM launches by grid (shape)"; python "$ROOT/tools/prof_by_grid.py" "$DB" "$STEPS"; } > "$OUTF"
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_one.log | tail -1
