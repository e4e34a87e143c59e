#!/usr/bin/env bash
# phase ablations of smx_gemm (diagnostic build): ablate bits 1 = no epilogue, 2 = no MFMA, 4 = no global loads
cd /root/repo
export SMX_LIB=${SMX_LIB:-/root/repo/summarymixing_amd/libsmx_diag.so}
SHAPES=${SHAPES:-"NTln 64000 1024 256;NNlnb 64000 1024 256;NT 64000 256 1024;NNag 64000 256 1024;NTlnc 64000 256 256;NN 64000 2048 512;NTres 64000 2048 512"}
IFS=';' read -ra SH <<< "$SHAPES"
for shape in "${SH[@]}"; do
  for ab in ${ABL:-0 1 2 4 6 3 5 7}; do
    echo -n "ablate=$ab  "
    SMX_GEMM_ABLATE=$ab python tools/one_gemm.py $shape 2>&1 | tail -1
  done
done
