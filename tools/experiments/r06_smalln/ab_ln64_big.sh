#!/usr/bin/env bash
# the 64-row LayerNorm tile above one round of 128-row tiles (N > 65 536 rows): SMX_LN_TILE64 = 0 (128 rows) / 1 (cost rule)
cd "$(dirname "$0")/../../.." || exit 1
one() { python bench.py "$@" --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for b in 136 160 200 256; do for m in 0 1; do echo "B=$b SMX_LN_TILE64=$m $(SMX_LN_TILE64=$m one --batch $b)"; done; done; done
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 %8.3f ms' % d['ms_per_step'])"
