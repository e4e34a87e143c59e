#!/usr/bin/env bash
# LayerNorm fusion at d_model = 512 (the row-complete 128 x 512 tile): both directions / forward only / backward only / none, per config
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$1 | $2 |" $(env $1 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
for m in 1 fwd bwd 0; do run SMX_LN_FUSE=$m "--config c2a"; done
for m in 1 fwd bwd 0; do run SMX_LN_FUSE=$m "--config c4"; done
for m in 1 0; do run SMX_LN_FUSE=$m "--config c5 --steps 6"; done
for m in 1 0; do run SMX_LN_FUSE=$m "--config c2a --mode forward"; done
done
