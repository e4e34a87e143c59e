#!/usr/bin/env bash
# the persistent 256 x 256 LDS-DMA GEMM (libsmx_pg.so = build.sh with SMX_VARIANT=pg SMX_CXXFLAGS=-DSMX_PGEMM_BUILD) against the
# tiled kernels, with and without the ping-pong issue order
cd /root/repo
SH=("NTres 64000 2048 512" "NN 64000 2048 512" "NTres 64000 1024 512" "NN 64000 1024 512" "NT 64000 512 512 plain" "NN 64000 512 512" "NTres 64000 1024 256" "NN 64000 1024 256" "NT 64000 512 2048" "NNag 64000 512 2048")
for mode in "libsmx.so 0 0" "libsmx_pg.so 1 0" "libsmx_pg.so 1 1"; do
  set -- $mode
  echo "== $1 SMX_PGEMM=$2 SMX_PGEMM_PP=$3"
  for shape in "${SH[@]}"; do
    SMX_LIB=/root/repo/summarymixing_amd/$1 SMX_PGEMM=$2 SMX_PGEMM_PP=$3 python tools/one_gemm.py $shape 2>&1 | tail -1
  done
done
