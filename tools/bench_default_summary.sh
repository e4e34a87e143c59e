#!/usr/bin/env bash
# The driver's default bench line + a short summary (value, roofline of the dominant kernel, the extra points with their roofline_step).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06; mkdir -p $O
SECONDS=0; python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "wall ${SECONDS} s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_default.json").read().strip().splitlines()[-1])
print(f"value {d['value']:.0f} frames/s  {d['ms_per_step']:.3f} ms/step  roofline frac {d['roofline']['frac']:.3f} [{d['roofline']['kernel'][:64]}]  cpu {d['cpu_baseline']['value']:.0f}")
print("roofline_step", {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline_step"].items() if k in ("ms_per_step", "launches", "frac_hbm", "frac_mfma")})
for p in d["extra_points"]:
    rs = p.get("roofline_step") or {}
    print(f"  {p.get('point', '')[:70]:70s} {p.get('ms_per_step', 0):8.3f} ms  frac_hbm {rs.get('frac_hbm')}  launches {rs.get('launches')}  {p.get('error', '')}")
PY
