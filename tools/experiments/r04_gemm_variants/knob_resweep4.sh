cd /root/repo
run() { echo "$2 $1" $(env $1 python bench.py ${2:-} --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
for c in "" "--config c2a" "--config c4" "--batch 64" "--batch 32" "--config c2a --batch 10 --frames 375" "--batch 1"; do
run X=0 "$c"; run SMX_WGRAD_STREAM=0 "$c"
done
done
