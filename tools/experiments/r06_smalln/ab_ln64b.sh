#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for bs in 36 48 64 68 72 96 128 144 160; do
  echo "C2b B=$bs: SMX_LN_TILE64=0 $(SMX_LN_TILE64=0 run --batch $bs) | =2 $(SMX_LN_TILE64=2 run --batch $bs) | auto $(run --batch $bs)"
done
