#!/usr/bin/env bash
# The four small/mid-N points of the round-5 review (item 1): recipe batch, 4 fused micro-batches, C2b B = 64, one utterance.  usage: bash tools/r06_points.sh <tag>
cd "$(dirname "$0")/.." || exit 1
T="${1:-x}"; O=gpurun_out/r06/points_$T; mkdir -p $O
B="--no-cpu-baseline --no-extra-points --no-roofline"
python bench.py --config c2a --batch 10 --frames 375 --steps 30 --warmup 5 $B 2>/dev/null | tail -1 > $O/recipe.json
python bench.py --config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused --steps 20 --warmup 5 $B 2>/dev/null | tail -1 > $O/recipe_accum4.json
python bench.py --batch 64 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 > $O/b64.json
python bench.py --batch 1 --steps 30 --warmup 5 $B 2>/dev/null | tail -1 > $O/b1.json
[[ "${FULL:-0}" == "1" ]] && python bench.py --steps 20 --warmup 5 $B 2>/dev/null | tail -1 > $O/default.json
[[ "${FULL:-0}" == "1" ]] && python bench.py --config c2a --steps 10 --warmup 3 $B 2>/dev/null | tail -1 > $O/c2a.json
for f in $O/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read()); print('%-22s %8.3f ms  %10.0f frames/s' % ('$(basename $f .json)', d['ms_per_step'], d['value']))"; done | tee $O/summary.txt
