cd /root/repo
B="--no-cpu-baseline --no-extra-points --no-roofline"
for lib in libsmx.so libsmx_wide192.so libsmx_wide128.so; do for bs in 64 48 32; do
 r=$(SMX_LIB=summarymixing_amd/$lib python bench.py --batch $bs --steps 15 --warmup 4 $B 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
 echo "$lib B=$bs $r"; done; done
