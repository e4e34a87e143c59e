#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
for v in "" _ns2 _ns4; do
  echo "## libsmx$v.so" 
  SMX_LIB=summarymixing_amd/libsmx$v.so python tools/experiments/r06_smalln/probe.py 2>&1 | grep -v amdgpu.ids
done > $O/probe_splitk.txt
for v in "" _ns2 _ns4; do
  echo "## libsmx$v.so"
  N=3750 SMX_LIB=summarymixing_amd/libsmx$v.so python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep "K=  512\|K= 2048"
done > $O/probe_ns_gemm_bench.txt
N=500 D=256 python tools/experiments/r06_smalln/probe.py 2>&1 | grep -v amdgpu.ids > $O/probe_splitk_b1.txt
cat $O/probe_splitk.txt $O/probe_ns_gemm_bench.txt $O/probe_splitk_b1.txt
