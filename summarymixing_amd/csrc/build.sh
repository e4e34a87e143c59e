#!/usr/bin/env bash
# Build libsmx.so (the C-ABI shared library of include/smx.h) for gfx950, in-tree.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# SMX_DIAG=1 bash build.sh: the diagnostic build (libsmx_diag.so, objects in obj_diag/) with the phase-ablation branches of the
# GEMM / grouped-wgrad / rolling-conv kernels compiled in (-DSMX_DIAG; tools/tn_ablate.py and the SMX_*_ABLATE variables).  The
# product library contains none of them.  SMX_LIB=<path> makes summarymixing_amd._lib load another build.
DIAG="${SMX_DIAG:-0}"
if [[ "$DIAG" == "1" ]]; then OUT="${HERE}/../libsmx_diag.so"; OBJ="${HERE}/obj_diag"; else OUT="${HERE}/../libsmx.so"; OBJ="${HERE}/obj"; fi
# SMX_VARIANT=<name> SMX_CXXFLAGS="-D..." bash build.sh: an A/B build (libsmx_<name>.so, objects in obj_<name>/; both git-ignored)
VARIANT="${SMX_VARIANT:-}"
if [[ -n "$VARIANT" ]]; then OUT="${HERE}/../libsmx_${VARIANT}.so"; OBJ="${HERE}/obj_${VARIANT}"; fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: the SLP vectoriser turns adjacent fp32 operations into v_pk_{add,mul,fma}_f32; beside MFMAs a v_pk_fma_f32
# costs ~14 cycles of issue against ~4 for a v_fma_f32 (tools/experiments/mfma_valu_probe.hip); round 2 found the same in the
# depthwise-conv FMA chains.  A/B on one box, whole library: C2b step 19.19 / 19.20 -> 19.04 / 19.03 ms.
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -fno-slp-vectorize -Wall -Wno-unused-function)
[[ "$DIAG" == "1" ]] && FLAGS+=(-DSMX_DIAG)
# shellcheck disable=SC2206
[[ -n "${SMX_CXXFLAGS:-}" ]] && FLAGS+=(${SMX_CXXFLAGS})
mkdir -p "$OBJ"
pids=()
for f in capi gemm gemm_ln256 gemm_ln256r64 gemm_ln512 gemm_panel gemm_panel_bwd rowwise slab_epilogue dwconv frontend ctc reduce wgrad_group; do
  src="${HERE}/${f}.hip"; obj="${OBJ}/${f}.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "${HERE}/smx_common.h" -nt "$obj" || "${HERE}/gemm_common.h" -nt "$obj" || "${HERE}/gemm_kernel.h" -nt "$obj" || "${HERE}/gemm_panel.h" -nt "$obj" || "${HERE}/dwconv_roll.h" -nt "$obj" || "${HERE}/build.sh" -nt "$obj" || "${HERE}/../../include/smx.h" -nt "$obj" ]]; then
    extra=()
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJ}"/{capi,gemm,gemm_ln256,gemm_ln256r64,gemm_ln512,gemm_panel,gemm_panel_bwd,rowwise,slab_epilogue,dwconv,frontend,ctc,reduce,wgrad_group}.o
echo "built $OUT"
