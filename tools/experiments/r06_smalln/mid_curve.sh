#!/usr/bin/env bash
# step time over the batch size between the whole-round sweet spots (C2b, T = 500), and per-kernel profiles of two in-between points
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/midcurve; mkdir -p $O
one() { python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %8.3f ms  %10.0f frames/s' % ('$*', d['ms_per_step'], d['value']))"; }
for b in 48 56 64 68 72 80 88 96 104 112 120 128 136 144 160; do one --batch $b; done > $O/curve.txt
cat $O/curve.txt
for b in 72 96; do
  bash tools/prof_one.sh $O/step_c2b_b${b}.txt 9 "rocprofv3 --kernel-trace --stats -- python bench.py --batch $b --steps 6 --warmup 2 (+1 capture)" python /root/repo/bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points
done
