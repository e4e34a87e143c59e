#!/usr/bin/env bash
# Re-measure everything profiles/ holds for this round (run on the GPU box through gpurun; outputs land in
# gpurun_out/refresh/, copy them into profiles/ afterwards):  gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r03 v2'
set -uo pipefail
R="${1:-r06}"; TAG="${2:-vX}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/refresh"; mkdir -p "$OUT"
cd "$ROOT"
python bench.py 2>/dev/null | tail -1 > "$OUT/${R}_bench_default.json"
python bench.py --batch 64 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${R}_bench_b64.json"
python bench.py --batch 32 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/${R}_bench_b32.json"
python bench.py --config c2a --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${R}_bench_c2a.json"
python bench.py --config c2a --batch 10 --frames 375 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/${R}_bench_c2a_recipe_batch.json"
python bench.py --mode forward --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${R}_bench_fwd.json"
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${R}_bench_c4.json"
python bench.py --config c5 --steps 5 --warmup 2 2>/dev/null | tail -1 > "$OUT/${R}_bench_c5.json"
python bench.py --dynchunk 8,2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/${R}_bench_dynchunk_8_2.json"
python bench.py --dynchunk 16 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/${R}_bench_dynchunk_16.json"
SMX_RESIDUAL=bf16 python bench.py --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 > "$OUT/${R}_bench_bf16_stream.json"
python tools/frontend_bench.py "$(python -c "import json;print(json.load(open('$OUT/${R}_bench_default.json'))['ms_per_step'])")" > "$OUT/${R}_frontend.txt" 2>/dev/null
prof() {  # name, header, command...
  local name="$1" hdr="$2"; shift 2
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_refresh && rocprofv3 --kernel-trace --stats -d /tmp/prof_refresh -o p -- "$@" > /dev/null 2>&1)
  local DB; DB=$(find /tmp/prof_refresh -name '*.db' | head -1)
  { echo "# $hdr"; python tools/prof_summary.py "$DB" "${STEPS:-1}"; echo; echo "## GEMM launches by grid (shape)"; python tools/prof_by_grid.py "$DB" "${STEPS:-1}"; } > "$OUT/$name"
}
STEPS=7 prof "${R}_step_c2b_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline   (7 steps incl. warmup; C2b, B=128 x T=500, bf16, dropout 0.15)" python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
STEPS=7 prof "${R}_step_c4_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline   (7 steps; C4 Branchformer, B=128 x T=250)" python "$ROOT/bench.py" --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
STEPS=7 prof "${R}_step_c2a_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --config c2a --steps 5 --warmup 2 --no-cpu-baseline --no-roofline   (7 steps; C2a Conformer d=512 f=2048, B=128 x T=500)" python "$ROOT/bench.py" --config c2a --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
STEPS=7 prof "${R}_fwd_c5_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline   (7 forward passes; C5 12-layer stack, 8 x 30000 x 512)" python "$ROOT/bench.py" --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
STEPS=25 prof "${R}_step_c2a_recipe_batch_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --config c2a --batch 10 --frames 375 --steps 20 --warmup 5   (25 steps; the recipe's batch of 150 s: 3750 frames, hipGraph replay)" python "$ROOT/bench.py" --config c2a --batch 10 --frames 375 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline
STEPS=7 prof "${R}_step_c2b_dynchunk_8_2_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --dynchunk 8,2 --steps 5 --warmup 2   (7 steps; C2b with DynChunk chunk 8, left context 2 chunks)" python "$ROOT/bench.py" --dynchunk 8,2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
STEPS=1 prof "${R}_frontend_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python tools/frontend_bench.py   (waveform -> fbank -> InputNormalization -> CNN fwd / fwd+bwd at B = 128 x 20 s; per-call averages are the figures, the 'step' total covers all timed repetitions)" python "$ROOT/tools/frontend_bench.py"
STEPS=25 prof "${R}_step_c2b_b1_${TAG}.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --batch 1 --steps 20 --warmup 5   (25 steps; C2b, ONE utterance of 500 frames, hipGraph replay)" python "$ROOT/bench.py" --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points
python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 > "$OUT/${R}_bench_b1.json"
STEPS=9 prof "${R}_step_c2b_b64_v2.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --batch 64 --steps 6 --warmup 2   (9 steps incl. the capture; C2b, B = 64 x T = 500: every launch a whole round)" python "$ROOT/bench.py" --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points
STEPS=9 prof "${R}_step_c2b_b72_v1.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --batch 72 --steps 6 --warmup 2   (9 steps incl. the capture; C2b, B = 72 x T = 500: just behind the whole-round sweet spot)" python "$ROOT/bench.py" --batch 72 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-points
{ for b in 48 56 64 68 72 80 96 112 128 136 160; do python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-points 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2b B = %3d x T = 500  %8.3f ms  %10.0f frames/s' % ($b, d['ms_per_step'], d['value']))"; done; } > "$OUT/${R}_batch_curve.txt"
python bench.py --config c2a --batch 10 --frames 375 --grad-accum 4 --accum fused --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > "$OUT/${R}_bench_c2a_recipe_accum4_fused.json"
python tools/blaslt_plus_epilogue.py > "$OUT/${R}_blaslt_plus_epilogue.txt" 2>/dev/null
D=512 F=2048 python tools/blaslt_plus_epilogue.py > "$OUT/${R}_blaslt_plus_epilogue_d512.txt" 2>/dev/null
# the panel-resident GEMM: isolated against the tiled kernel, and the steps with it off / on (same box)
D=256 F=1024 python tools/panel_bench.py 2>/dev/null | grep -v amdgpu > "$OUT/${R}_panel_bench_d256.txt"
D=512 F=2048 python tools/panel_bench.py 2>/dev/null | grep -v amdgpu > "$OUT/${R}_panel_bench_d512.txt"
D=512 F=3072 python tools/panel_bench.py 2>/dev/null | grep -v amdgpu > "$OUT/${R}_panel_bench_d512_f3072.txt"
bash tools/experiments/ab_panel.sh > "$OUT/${R}_ab_panel.txt" 2>&1
STEPS=8 prof "${R}_wgrad_group_isolated.txt" "rocprofv3 --kernel-trace --stats -- python tools/one_wgroup.py 64000 layer   (8 launches: the 8 weight gradients of a C2b layer, 64000 frames, isolated back to back; algorithmic 983 + 1.4 MB per launch)" python "$ROOT/tools/one_wgroup.py" 64000 layer
STEPS=8 prof "${R}_wgrad_group_one_1024x256.txt" "rocprofv3 --kernel-trace --stats -- python tools/one_wgroup.py 64000 one   (8 launches: dW(1024x256) alone over 64000 frames; algorithmic 164.9 MB per launch)" python "$ROOT/tools/one_wgroup.py" 64000 one
# counter passes last and only on request (PMC=1): after them the box has been seen to lose its device for the next process
if [[ "${PMC:-0}" == "1" ]]; then
  bash tools/pmc_traffic.sh > "$OUT/${R}_pmc_traffic.txt" 2>&1
  bash tools/pmc_wgroup.sh layer > "$OUT/${R}_pmc_wgrad_group.txt" 2>&1
  bash tools/pmc_panel.sh > "$OUT/${R}_pmc_panel.txt" 2>&1
fi
ls -la "$OUT"
