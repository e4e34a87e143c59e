#!/usr/bin/env bash
# HBM traffic of the two roofline kernels from the TCC fabric counters (separate --pmc passes, kernel-trace only).
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md
# §HBM): the summary prints raw and corrected (x2) read bytes per launch.
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp; export TMPDIR=/tmp
run() {  # name, kernel substring, command...
  local name="$1" pat="$2"; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- "$@" > /dev/null 2>&1 || true
  done
  python3 - "$name" "$pat" <<'PY'
import csv, glob, sys, collections
name, pat = sys.argv[1], sys.argv[2]
if name.startswith("@"):                       # "@ suffix": the launch's own in-step name (written by tools/gemm_bench.py) + suffix,
    try:                                       # and its algorithmic bytes from the SAME model bench.py uses (ops.gemm's record)
        rec = open("/tmp/pmc_name.txt").read().split("\n")
        alg = f"; algorithmic {float(rec[1]) / 1e6:.1f} MB]" if len(rec) > 1 and rec[1].strip() else "]"
        name = rec[0].strip() + " " + name[1:].strip().rstrip("]") + alg
    except (OSError, ValueError):
        pass
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
run "pool bf16 (8,30000,512) masked_sum_stage1 [algorithmic 246.0 MB]" masked_sum_stage1 python3 "$ROOT/tools/one_pool.py" bf16
run "pool f32  (8,30000,512) masked_sum_stage1 [algorithmic 491.8 MB]" masked_sum_stage1 python3 "$ROOT/tools/one_pool.py" f32
run "gemm NT bf16 (64000x256)x(256x1024)+bias+swish+Z = bench.py roofline kernel [algorithmic 295.4 MB: 33.3 read, 262.1 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NT 64000 256 1024
run "gemm NT bf16 (32000x256)x(256x1024)+bias+swish+Z [algorithmic 147.9 MB: 16.9 read, 131.1 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NT 32000 256 1024
run "gemm NT bf16 (32000x512)x(512x2048)+bias+swish+Z [algorithmic 297.0 MB: 34.9 read, 262.1 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NT 32000 512 2048
run "gemm NN bf16 (64000x256)x(256x1024) +actgrad(z)+drop [algorithmic 295.4 MB: 164.3 read, 131.1 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NNag 64000 256 1024
run "gemm NN bf16 64000x1024 . 1024x256 plain dgrad, no LayerNorm epilogue [algorithmic 164.4 MB: 131.6 read, 32.8 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NN 64000 1024 256
run "gemm NT bf16 64000x1024 . 1024x256 +bias+res+drop, no LayerNorm epilogue [algorithmic 197.1 MB: 164.4 read, 32.8 write]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTres 64000 1024 256
run "dwconv_bwd (128,500,256) k=31 [algorithmic 163.8 MB: 98.3 read, 65.5 write]" dwconv_rolls_bwd python3 "$ROOT/tools/one_dwconv.py"
run "dwconv_fwd (128,500,256) k=31 [algorithmic 98.3 MB: 65.5 read, 32.8 write]" dwconv_rolls_fwd python3 "$ROOT/tools/one_dwconv.py"
# ---- round 3: the LayerNorm-fused instantiations on the float32 residual stream (names = the in-step names of bench.py) ----
run "@ [FFN down-projection -> norm1]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTln 64000 1024 256
run "@ [FFN down-projection -> norm2, fp32 LayerNorm output]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTln2 64000 1024 256
run "@ [the cell's merge, K = l + s]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTlnm 64000 512 256
run "@ [conv-module out-projection]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NTlnc 64000 256 256
run "@ [dgrad K=1024 + LayerNorm backward, fp32 ln_x]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NNlnb 64000 1024 256
run "@ [dgrad K=512 + LayerNorm backward, fp32 ln_x]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NNlnb 64000 512 256
run "@ [dgrad K=512 + LayerNorm backward + act-grad second output]" gemm_kernel python3 "$ROOT/tools/one_gemm.py" NNlnb3 64000 512 256
