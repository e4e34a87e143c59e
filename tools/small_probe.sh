#!/usr/bin/env bash
# Round 6: where do the small-grid (recipe batch, 3750 frames) GEMM launches spend their time?  Isolated timings + per-wave stamps.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/small_probe; mkdir -p $O
N=3750 python tools/gemm_bench.py > $O/gemm_bench_3750.txt 2>&1
for spec in "NT 2048 512 plain" "NT 512 2048" "NT 512 1024" "NT 512 512 plain" "NN 2048 512" "NN 512 2048" "NN 512 512" "NNag 512 2048" "NTres 2048 512"; do
  set -- $spec
  N=3750 SMX_LIB=summarymixing_amd/libsmx_diag.so python tools/gemm_stamps.py "$@" >> $O/stamps_3750.txt 2>&1
done
python bench.py --config c2a --batch 10 --frames 375 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_recipe.json 2>$O/bench_recipe.err
python bench.py --config c2b --batch 64 --frames 500 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_b64.json 2>$O/bench_b64.err
python bench.py --config c2b --batch 1 --frames 500 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-points > $O/bench_b1.json 2>$O/bench_b1.err
tail -n 3 $O/*.json
