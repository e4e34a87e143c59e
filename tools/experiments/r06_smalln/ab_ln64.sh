#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for bs in 36 48 64 72 80 96 128; do
  echo "C2b B=$bs: 128-row tiles $(SMX_LN_TILE64_MAX_TILES=0 run --batch $bs) $(SMX_LN_TILE64_MAX_TILES=0 run --batch $bs) | 64-row tiles $(SMX_LN_TILE64_MAX_TILES=100000 run --batch $bs) $(SMX_LN_TILE64_MAX_TILES=100000 run --batch $bs)"
done
