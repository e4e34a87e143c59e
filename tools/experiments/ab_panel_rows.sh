#!/usr/bin/env bash
# where does the panel path start to pay?  C2b / C2a steps at smaller batches with the row threshold lowered
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "--batch 64" "--batch 32" "--batch 16" "--config c2a --batch 32" "--config c2a --batch 64"; do
  for p in 0 1 0 1; do
    echo -n "SMX_PANEL=$p (min rows 4096) bench.py $cfg : "
    SMX_PANEL=$p SMX_PANEL_MIN_ROWS=4096 python bench.py $cfg --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
