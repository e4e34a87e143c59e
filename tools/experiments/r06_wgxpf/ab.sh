#!/usr/bin/env bash
# A/B of the cross-barrier fragment prefetch in wgrad_group_kernel<32> (SMX_WG_XPF): isolated launches, stamps, parity tests, steps
cd "$(dirname "$0")/../../.." || exit 1
mkdir -p gpurun_out/wgxpf
O=gpurun_out/wgxpf
{
for rep in 1 2; do
  for lib in libsmx.so libsmx_xpf0.so; do
    echo "== $lib"
    SMX_LIB=summarymixing_amd/$lib python tools/one_wgroup.py 64000 layer 2
    SMX_LIB=summarymixing_amd/$lib python tools/one_wgroup.py 64000 c2a 2
    SMX_LIB=summarymixing_amd/$lib python tools/one_wgroup.py 16000 layer 1
  done
done
echo "== stamps (diag = xpf1)"
SMX_LIB=summarymixing_amd/libsmx_diag.so python tools/wgroup_stamps.py
SMX_WGROUP_ABLATE=2 SMX_LIB=summarymixing_amd/libsmx_diag.so python tools/wgroup_stamps.py
} > $O/ab.txt 2>&1
python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -m gpu -k "wgrad or gradient" -x > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for rep in 1 2; do
  for lib in libsmx.so libsmx_xpf0.so; do
    echo "== $lib" >> $O/steps.txt
    SMX_LIB=summarymixing_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points >> $O/steps.txt 2>&1
    SMX_LIB=summarymixing_amd/$lib python bench.py --config c2a --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points >> $O/steps.txt 2>&1
  done
done
cat $O/ab.txt
grep -o '"ms_per_step": [0-9.]*\|== .*' $O/steps.txt
