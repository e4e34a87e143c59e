#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for bs in 32 48 64 80 96; do
  echo "C2b B=$bs: default $(run --batch $bs) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --batch $bs) | MIN_ROWS=0 fwd only $(SMX_LN_FUSE=fwd SMX_LN_FUSE_MIN_ROWS=0 run --batch $bs) | MIN_ROWS=0 bwd only $(SMX_LN_FUSE=bwd SMX_LN_FUSE_MIN_ROWS=0 run --batch $bs)"
done
for bs in 32 64; do
  echo "C2a B=$bs: default $(run --config c2a --batch $bs) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --config c2a --batch $bs) | fwd only $(SMX_LN_FUSE=fwd SMX_LN_FUSE_MIN_ROWS=0 run --config c2a --batch $bs) | bwd only $(SMX_LN_FUSE=bwd SMX_LN_FUSE_MIN_ROWS=0 run --config c2a --batch $bs)"
done
