#!/usr/bin/env bash
# Build libsmx.so (the C-ABI shared library of include/smx.h) for gfx950, in-tree.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libsmx.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: the SLP vectoriser turns adjacent fp32 operations into v_pk_{add,mul,fma}_f32; beside MFMAs a v_pk_fma_f32
# costs ~14 cycles of issue against ~4 for a v_fma_f32 (tools/experiments/mfma_valu_probe.hip); round 2 found the same in the
# depthwise-conv FMA chains.  A/B on one box, whole library: C2b step 19.19 / 19.20 -> 19.04 / 19.03 ms.
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -fno-slp-vectorize -Wall -Wno-unused-function)
mkdir -p "${HERE}/obj"
pids=()
for f in capi gemm rowwise dwconv frontend ctc reduce wgrad_group; do
  src="${HERE}/${f}.hip"; obj="${HERE}/obj/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${HERE}/smx_common.h" -nt "$obj" || "${HERE}/gemm_common.h" -nt "$obj" || "${HERE}/dwconv_roll.h" -nt "$obj" || "${HERE}/build.sh" -nt "$obj" || "${HERE}/../../include/smx.h" -nt "$obj" ]]; then
    extra=()
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${HERE}"/obj/{capi,gemm,rowwise,dwconv,frontend,ctc,reduce,wgrad_group}.o
echo "built $OUT"
