cd /root/repo
run() { echo "$1" $(env $1 python bench.py ${2:-} --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
run "X=0"
run "SMX_REG_EPI=1"
run "SMX_REG_EPI=0"
run "SMX_EPI_SIMPLE=1"
run "SMX_EPI_SIMPLE=0"
run "SMX_WGROUP_BK=64"
run "SMX_WGROUP_PP=0"
run "SMX_WGROUP_PP=2"
run "SMX_WGROUP_BLOCKS=512"
run "SMX_POOL_BLOCKS=1024"
done
for rep in 1 2; do
run "X=0" "--config c2a"
run "SMX_WGROUP_BK=64" "--config c2a"
run "SMX_WGROUP_PP=0" "--config c2a"
run "SMX_WGROUP_BLOCKS=512" "--config c2a"
run "SMX_REG_EPI=1" "--config c2a"
done
