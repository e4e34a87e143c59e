cd /root/repo
run() { echo "$2 $1" $(env $1 python bench.py ${2:-} --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
for k in X=0 SMX_POOL_BLOCKS=256 SMX_POOL_BLOCKS=1024 SMX_POOL_BLOCKS=2048 SMX_WGRAD_STREAM=0 SMX_WGRAD_ASYNC=1 SMX_TN_DMA=0; do run $k ""; done
for k in X=0 SMX_T256=0 SMX_POOL_BLOCKS=1024 SMX_WGRAD_ASYNC=1; do run $k "--config c2a"; done
for k in X=0 SMX_WGRAD_ASYNC=1 SMX_WGRAD_ASYNC=0; do run $k "--batch 32"; done
for k in X=0 SMX_WGRAD_ASYNC=1; do run $k "--batch 64"; done
for k in X=0 SMX_T256=0 SMX_PREACT_LN=0 SMX_SPLIT_MERGE_DGRAD=0; do run $k "--config c4"; done
done
