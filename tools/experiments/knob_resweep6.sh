cd /root/repo
run() { echo "$2 $1" $(env $1 python bench.py ${2:-} --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'); }
for rep in 1 2; do
run X=0 "--batch 64 --graph on"; run X=0 "--batch 64 --graph off"
run X=0 "--batch 96 --graph on"; run X=0 "--batch 96 --graph off"
run X=0 "--batch 128 --graph on"; run X=0 "--batch 128 --graph off"
run X=0 "--config c2a --batch 64 --graph on"; run X=0 "--config c2a --batch 64 --graph off"
run SMX_DEFER_REDUCE=0 ""; run SMX_WGRAD_GROUP=0 ""
run X=0 "--mode forward --graph on"; run X=0 "--mode forward --graph off"
run X=0 "--config c5 --steps 6 --graph on"; run X=0 "--config c5 --steps 6 --graph off"
done
