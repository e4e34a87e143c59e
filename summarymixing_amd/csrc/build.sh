#!/usr/bin/env bash
# Build libsmx.so (the C-ABI shared library of include/smx.h) for gfx950, in-tree.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libsmx.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wall -Wno-unused-function)
mkdir -p "${HERE}/obj"
pids=()
for f in capi gemm rowwise dwconv frontend ctc reduce wgrad_group; do
  src="${HERE}/${f}.hip"; obj="${HERE}/obj/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${HERE}/smx_common.h" -nt "$obj" || "${HERE}/gemm_common.h" -nt "$obj" || "${HERE}/dwconv_roll.h" -nt "$obj" || "${HERE}/build.sh" -nt "$obj" || "${HERE}/../../include/smx.h" -nt "$obj" ]]; then
    extra=()
    # dwconv: the SLP vectoriser turns the register-window FMA chains into v_pk_fma_f32 + v_pk_mov + s_nop (measured slower)
    [[ "$f" == dwconv ]] && extra=(-fno-slp-vectorize)
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${HERE}"/obj/{capi,gemm,rowwise,dwconv,frontend,ctc,reduce,wgrad_group}.o
echo "built $OUT"
