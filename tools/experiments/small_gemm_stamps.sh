cd /root/repo
export SMX_LIB=summarymixing_amd/libsmx_diag.so
for args in "NT 512 512" "NN 512 512" "NT 2048 512" "NN 2048 512" "NT 512 2048" "NN 512 2048"; do
  N=3750 python tools/gemm_stamps.py $args 2>&1 | grep -v amdgpu.ids
done
unset SMX_LIB
for sh in "NT 3750 512 512 bias" "NN 3750 512 512" "NT 3750 2048 512 bias" "NN 3750 2048 512" "NT 3750 512 2048" "NN 3750 512 2048"; do
  python tools/one_gemm.py $sh 2>&1 | grep -v amdgpu.ids
done
