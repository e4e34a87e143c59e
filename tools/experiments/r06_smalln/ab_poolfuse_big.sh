#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
run() { python bench.py "$@" --steps 15 --warmup 4 --no-cpu-baseline --no-extra-points --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for cfgargs in "--config c2b" "--config c2b --batch 64" "--config c2a" "--config c4"; do
  echo "$cfgargs: default $(run $cfgargs) $(run $cfgargs) | SMX_POOL_FUSE_MAX_ROWS=1000000 $(SMX_POOL_FUSE_MAX_ROWS=1000000 run $cfgargs) $(SMX_POOL_FUSE_MAX_ROWS=1000000 run $cfgargs)"
done
