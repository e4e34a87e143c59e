#!/usr/bin/env bash
# HBM traffic of the panel-resident GEMM from the TCC fabric counters (separate --pmc passes, kernel-trace only; see pmc_traffic.sh
# for the unit / gfx950 corrections).   gpurun -- 'bash tools/pmc_panel.sh > gpurun_out/r05_pmc_panel.txt'
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
run() {  # name, kernel substring, command...
  local name="$1" pat="$2"; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- "$@" > /dev/null 2>&1 || true
  done
  python3 - "$name" "$pat" <<'PY'
import csv, glob, sys
name, pat = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for fn in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    out[c] = (sum(vals) / len(vals) if vals else float("nan"), len(vals))
f, w = out["FETCH_SIZE"][0] * 1024, out["WRITE_SIZE"][0] * 1024
print(f"{name}: launches={out['FETCH_SIZE'][1]} FETCH_SIZE={f/1e6:.1f} MB (x2 corrected {2*f/1e6:.1f} MB)  WRITE_SIZE={w/1e6:.1f} MB  "
      f"traffic(corrected)={(2*f+w)/1e6:.1f} MB per launch")
PY
}
run "gemm panel bf16 (64000x256)x(256x1024) +act+Z+drop [algorithmic 295.4 MB: 33.3 read, 262.1 write]" gemm_panel_kernel python3 "$ROOT/tools/one_panel.py" fwd 64000 256 1024
run "gemm panel bf16 (64000x256)x(256x1024) +actgrad(z)+drop [algorithmic 295.4 MB: 164.3 read, 131.1 write]" gemm_panel_kernel python3 "$ROOT/tools/one_panel.py" ag 64000 256 1024
run "gemm panel bf16 (64000x512)x(512x2048) +act+Z+drop [algorithmic 591.9 MB: 67.6 read, 524.3 write]" gemm_panel_kernel python3 "$ROOT/tools/one_panel.py" fwd 64000 512 2048
run "gemm panel bf16 (64000x512)x(512x2048) +actgrad(z)+drop [algorithmic 591.9 MB: 329.8 read, 262.1 write]" gemm_panel_kernel python3 "$ROOT/tools/one_panel.py" ag 64000 512 2048
