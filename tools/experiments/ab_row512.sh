#!/usr/bin/env bash
# the row-complete 128 x 512 tile (LayerNorm fused, one workgroup per CU) against what it replaces, 64 000 frames, d_model = 512
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== LayerNorm-fused launches on the 128 x 512 tile"
for shape in "NTln 64000 2048 512" "NTln2 64000 2048 512" "NNlnb 64000 2048 512" "NTlnm 64000 1024 512" "NTlnc 64000 512 512" "NNlnb 64000 1024 512" "NNlnb 64000 512 512" "NNlnb3 64000 512 512"; do
  python tools/one_gemm.py $shape 2>&1 | tail -1
done
echo "== plain epilogues: SMX_T256=0 (128 x 256 tile) / 1 (256 x 256 where eligible) / 3 (128 x 512)"
for t in 0 1 3; do
  echo "-- SMX_T256=$t"
  for shape in "NTres 64000 2048 512" "NN 64000 2048 512" "NT 64000 2048 512 plain" "NTres 64000 1024 512" "NN 64000 1024 512" "NN 64000 512 512"; do
    SMX_T256=$t python tools/one_gemm.py $shape 2>&1 | tail -1
  done
done
echo "== yardstick"
D=512 F=2048 python tools/blaslt_plus_epilogue.py
echo "== ... with the LayerNorm as a separate launch (round 4)"
SMX_LN_FUSE=0 D=512 F=2048 python tools/blaslt_plus_epilogue.py | tail -2
