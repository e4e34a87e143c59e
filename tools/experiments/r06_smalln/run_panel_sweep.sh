#!/usr/bin/env bash
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r06; mkdir -p $O
for D in 512 256; do for r in 0 128 64 32; do D=$D SMX_PANEL_ROWS=$r python tools/experiments/r06_smalln/panel_sweep.py 2>&1 | grep -v amdgpu.ids; done; done > $O/panel_sweep.txt
python - <<'PY'
import collections,re
t=collections.defaultdict(dict)
for l in open('gpurun_out/r06/panel_sweep.txt'):
    m=re.match(r"d=(\d+) N=\s*(\d+) M=\s*(\d+) (\S+)\s+fwd\+swish\+Z\+drop\s+([\d.]+)\s+dgrad-plain\s+([\d.]+)",l)
    if m: t[(int(m[1]),int(m[2]),int(m[3]))][m[4]]=(float(m[5]),float(m[6]))
print("# d N M | fwd: tiled p128 p64 p32 | dgrad-plain: tiled p128 p64 p32")
for k in sorted(t):
    v=t[k]; g=lambda n,i: f"{v[n][i]:6.1f}" if n in v else "   -  "
    print(f"{k[0]:4d} {k[1]:6d} {k[2]:5d} | "+" ".join(g(n,0) for n in ("tiled","p128","p64","p32"))+" | "+" ".join(g(n,1) for n in ("tiled","p128","p64","p32")))
PY
