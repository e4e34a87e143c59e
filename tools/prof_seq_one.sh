#!/usr/bin/env bash
# bash tools/prof_seq_one.sh <out prefix> <command...>: rocprofv3 kernel trace -> <prefix>_summary.txt (per kernel) + <prefix>_seq.txt (every launch of the last step)
set -uo pipefail
P="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$(dirname "$P")"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_sq && rocprofv3 --kernel-trace --stats -d /tmp/prof_sq -o p -- "$@" > /tmp/prof_sq.log 2>&1)
DB=$(find /tmp/prof_sq -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- $*"; python "$ROOT/tools/prof_summary.py" "$DB" 1; echo; echo "## GEMM launches by grid (shape)"; python "$ROOT/tools/prof_by_grid.py" "$DB" 1; } > "${P}_summary.txt"
python "$ROOT/tools/prof_fullseq.py" "$DB" > "${P}_seq.txt"
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_sq.log | tail -1
