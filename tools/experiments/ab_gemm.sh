#!/usr/bin/env bash
# A/B of two libsmx builds on the hot GEMM shapes: ab_gemm.sh libA.so libB.so
cd /root/repo
for lib in "$@"; do
  echo "== $lib"
  for shape in "NTln 64000 1024 256" "NNlnb 64000 1024 256" "NTlnc 64000 256 256" "NTlnm 64000 256 256" "NNlnb3 64000 256 256" "NN 64000 1024 256" "NN 64000 2048 512" "NTres 64000 2048 512" "NT 64000 512 512 plain" "NN 64000 512 512" "NN 64000 1024 512"; do
    SMX_LIB=/root/repo/summarymixing_amd/$lib python tools/one_gemm.py $shape 2>&1 | tail -1
  done
done
