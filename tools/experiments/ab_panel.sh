#!/usr/bin/env bash
# same-box A/B of the panel-resident GEMM in the training steps:  gpurun -- 'bash tools/experiments/ab_panel.sh'
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "" "--config c2a" "--config c4" "--config c5 --steps 5 --warmup 2"; do
  for p in 0 1 0 1; do
    echo -n "SMX_PANEL=$p bench.py $cfg : "
    SMX_PANEL=$p python bench.py $cfg --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'])"
  done
done
