#!/usr/bin/env bash
# A/B of read-once library knobs on the default bench step (C2b, B = 128): bash tools/knob_sweep.sh "VAR=val ..." "VAR=val" ...
# (each argument is one environment; prints ms per step)
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), '$*')"; }
for e in "$@"; do run $e; done
