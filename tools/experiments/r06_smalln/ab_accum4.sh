#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
B="--config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points --no-roofline"
run() { python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "default                      $(run) $(run)"
echo "SMX_PANEL_MIN_ROWS=12288     $(SMX_PANEL_MIN_ROWS=12288 run) $(SMX_PANEL_MIN_ROWS=12288 run)"
echo "SMX_POOL_FUSE=0              $(SMX_POOL_FUSE=0 run) $(SMX_POOL_FUSE=0 run)"
echo "both off                     $(SMX_POOL_FUSE=0 SMX_PANEL_MIN_ROWS=12288 run) $(SMX_POOL_FUSE=0 SMX_PANEL_MIN_ROWS=12288 run)"
echo "SMX_PANEL_ROWS=128           $(SMX_PANEL_ROWS=128 run) $(SMX_PANEL_ROWS=128 run)"
