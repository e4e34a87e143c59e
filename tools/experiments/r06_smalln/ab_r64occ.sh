#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
B="--no-cpu-baseline --no-extra-points --no-roofline --steps 15 --warmup 4"
for bs in 48 64 72 80 96; do
 line="C2b B=$bs:"
 for lib in libsmx.so libsmx_r64o3.so; do
  r=$(SMX_LIB=summarymixing_amd/$lib python bench.py --batch $bs $B 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  line="$line $lib $r |"
 done
 echo "$line"
done
