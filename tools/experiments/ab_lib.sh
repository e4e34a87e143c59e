#!/usr/bin/env bash
# same-box A/B of two builds of the library on bench configs: bash tools/experiments/ab_lib.sh <variant name> ["bench args" ...]
# (the variant: SMX_VARIANT=<name> SMX_CXXFLAGS="-D..." bash summarymixing_amd/csrc/build.sh -> summarymixing_amd/libsmx_<name>.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
run() { echo "$1 | $2 |" $(env $1 python bench.py $2 --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
[[ $# -eq 0 ]] && set -- ""
for rep in 1 2 3; do
  for a in "$@"; do
    run "SMX_LIB=$PWD/summarymixing_amd/libsmx.so" "$a"
    run "SMX_LIB=$PWD/summarymixing_amd/libsmx_$V.so" "$a"
  done
done
