cd /root/repo
for v in 1 0; do
  export SMX_LN_PAIR=$v
  bash tools/prof_one.sh gpurun_out/pair_c2a_$v.txt 7 "c2a SMX_LN_PAIR=$v" python /root/repo/bench.py --config c2a --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
done
