#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for bs in 64 68 72 80 88 96 104 128 144; do
  echo "C2b B=$bs: forced 128-row panels $(SMX_PANEL_ROWS=128 run --batch $bs) | auto $(run --batch $bs) $(run --batch $bs)"
done
for bs in 72 80; do echo "C2a B=$bs: forced 128 $(SMX_PANEL_ROWS=128 run --config c2a --batch $bs) | auto $(run --config c2a --batch $bs)"; done
