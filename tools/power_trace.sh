#!/usr/bin/env bash
# Clock / power evidence for the C2b step and for its GEMM kernels in isolation.
#   gpurun --timeout 900 -- 'bash tools/power_trace.sh'   -> gpurun_out/power/*.csv + summary.txt (copy into profiles/r05_power_trace.txt)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/power"; mkdir -p "$OUT"
cd "$ROOT"
gcc -O2 tools/power_trace.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o /tmp/power_trace || exit 1
python -c "import torch; torch.zeros(1).cuda()" > /dev/null 2>&1     # page the image in before anything is timed
trace() {  # name, seconds, command...
  local name="$1" secs="$2"; shift 2
  /tmp/power_trace "$secs" > "$OUT/$name.csv" 2> "$OUT/$name.err" &
  local pid=$!
  sleep 1
  "$@" > "$OUT/$name.log" 2>&1
  wait $pid
  { python tools/power_summary.py "$OUT/$name.csv" "$name"; grep -h "us per launch\|ms_per_step" "$OUT/$name.log" | cut -c1-400 | sed 's/^/  | /'; echo; } >> "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -30 > "$OUT/smi_idle.txt"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|cap" >> "$OUT/smi_idle.txt"
S="${SECS:-6}"
trace step_c2b 22 python bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points
trace chain_ffn_c2b $((S + 8)) python tools/loop_gemm.py "$S" "NT 64000 256 1024" "NTln 64000 1024 256" "NNlnb 64000 1024 256" "NNag 64000 256 1024"
trace one_NTln_1024_256 $((S + 8)) python tools/loop_gemm.py "$S" "NTln 64000 1024 256"
trace one_NNag_256_1024 $((S + 8)) python tools/loop_gemm.py "$S" "NNag 64000 256 1024"
trace one_NT_256_1024 $((S + 8)) python tools/loop_gemm.py "$S" "NT 64000 256 1024"
trace one_NN_2048_512 $((S + 8)) python tools/loop_gemm.py "$S" "NN 64000 2048 512"
trace one_wgroup_layer $((S + 8)) python tools/one_wgroup.py 64000 layer "$S"
trace pool_c5 $((S + 8)) python tools/one_pool.py "$S"
if [[ -f summarymixing_amd/libsmx_diag.so ]]; then   # the phase ablations: what do the PARTS clock at?
  for ab in 2 4 1; do
    SMX_LIB=$ROOT/summarymixing_amd/libsmx_diag.so SMX_GEMM_ABLATE=$ab trace "ablate${ab}_NTln_1024_256" $((S + 8)) python tools/loop_gemm.py "$S" "NTln 64000 1024 256"
  done
fi
# per-kernel effective clock of the step: GRBM_GUI_ACTIVE / wall time (counters in their own pass, kernel-trace only)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/grbm_step && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/grbm_step -o p -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points > /tmp/grbm_step.log 2>&1)
{ echo "## GRBM_GUI_ACTIVE per kernel, C2b step (5 steps, profiled = kernels serialised)"; python tools/grbm_clock.py /tmp/grbm_step 14; echo; } >> "$OUT/summary.txt"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/grbm_one && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/grbm_one -o p -- python "$ROOT/tools/loop_gemm.py" 1 "NTln 64000 1024 256" > /tmp/grbm_one.log 2>&1)
{ echo "## GRBM_GUI_ACTIVE, NTln 64000 1024 256 alone"; python tools/grbm_clock.py /tmp/grbm_one 3; echo; } >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
