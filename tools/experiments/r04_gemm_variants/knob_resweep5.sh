cd /root/repo
run() { echo "$2 $1" $(env $1 python bench.py ${2:-} --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
for c in "--batch 1" "--config c2a --batch 10 --frames 375" "--batch 8" "--batch 16" "--config c2a --batch 1 --frames 375" "--config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused"; do
run X=0 "$c"; run SMX_WGRAD_ASYNC=0 "$c"; run SMX_WGRAD_STREAM=0 "$c"
done
done
