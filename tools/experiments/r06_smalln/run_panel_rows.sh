#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
{ for r in 128 64 32; do SMX_PANEL_ROWS=$r N=${N:-3750} D=${D:-512} python tools/experiments/r06_smalln/panel_rows.py 2>&1 | grep -v amdgpu.ids; done; } > $O/panel_rows_N${N:-3750}_D${D:-512}.txt
cat $O/panel_rows_N${N:-3750}_D${D:-512}.txt
