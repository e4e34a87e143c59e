#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for bs in 36 40 44 56; do
  echo "C2b B=$bs: default $(run --batch $bs) $(run --batch $bs) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --batch $bs) $(SMX_LN_FUSE_MIN_ROWS=0 run --batch $bs)"
done
for bs in 40 48 56; do
  echo "C2a B=$bs: default $(run --config c2a --batch $bs) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --config c2a --batch $bs)"
done
echo "C4 B=64: default $(run --config c4 --batch 64) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --config c4 --batch 64)"
echo "C4 B=96: default $(run --config c4 --batch 96) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --config c4 --batch 96)"
echo "recipe accum4: default $(run --config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused) | MIN_ROWS=0 $(SMX_LN_FUSE_MIN_ROWS=0 run --config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused)"
