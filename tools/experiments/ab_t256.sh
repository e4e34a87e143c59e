#!/usr/bin/env bash
# the 256 x 256 tile (one workgroup per CU, 128 x 128 per wave) against the 128 x 256 tile on the long-K shapes
cd /root/repo
SH=("NTres 64000 2048 512" "NN 64000 2048 512" "NTres 64000 1024 512" "NN 64000 1024 512" "NT 64000 512 512 plain" "NN 64000 512 512" "NTres 64000 1024 256" "NN 64000 1024 256" "NT 64000 512 2048" "NNag 64000 512 2048" "NT 240000 2048 512 plain")
for t in 0 1; do
  echo "== SMX_T256=$t"
  for shape in "${SH[@]}"; do
    SMX_T256=$t python tools/one_gemm.py $shape 2>&1 | tail -1
  done
done
