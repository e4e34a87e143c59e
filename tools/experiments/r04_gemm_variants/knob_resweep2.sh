cd /root/repo
run() { echo "$2 $1" $(env $1 python bench.py ${2:-} --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
for c in "--config c2a" "--config c4" "--config c5" "--mode forward" "--batch 64" "--config c2a --batch 10 --frames 375"; do
run "X=0" "$c"; run "SMX_REG_EPI=0" "$c"
done
done
for rep in 1 2; do
for b in 0 460 552 644; do run "SMX_WGROUP_BLOCKS=$b" "--config c2a"; done
for b in 0 512 768; do run "SMX_WGROUP_BLOCKS=$b" "--config c4"; done
done
