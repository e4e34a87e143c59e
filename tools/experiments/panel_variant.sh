#!/usr/bin/env bash
# A/B build of the panel GEMM only: tools/experiments/panel_variant.sh <name> "<-D flags>"  ->  summarymixing_amd/libsmx_<name>.so
# (every other object is copied from the product build; run with SMX_LIB=summarymixing_amd/libsmx_<name>.so)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
name="$1"; flags="${2:-}"
rm -rf "$HERE/summarymixing_amd/csrc/obj_$name"
cp -r "$HERE/summarymixing_amd/csrc/obj" "$HERE/summarymixing_amd/csrc/obj_$name"
rm -f "$HERE/summarymixing_amd/csrc/obj_$name"/gemm_panel*.o
touch -d '+1 second' "$HERE/summarymixing_amd/csrc/obj_$name"/*.o
SMX_VARIANT="$name" SMX_CXXFLAGS="$flags" bash "$HERE/summarymixing_amd/csrc/build.sh"
