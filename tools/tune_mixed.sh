#!/bin/bash
# A/B variants of the f32-MFMA S build at M > 16 (prefetch depth).  Usage: tune_mixed.sh build | run I A B --M 32
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("x_base:" "x_nofast:-DCMTFPLS_MIXED_FAST=0" "x_w2:-DCMTFPLS_MIXED_UN_WIDE=2")
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"; rm -f "$OUT"/libcmtfpls_x_*.so
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags "$SRC"/{runtime,sweeps,small,rank1,rank1_tensor,xcov,mttkrp,mixed}.hip -o "$OUT/libcmtfpls_$name.so" ) &
  done
  wait; ls "$OUT" | grep x_
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    echo "=== variant $name"
    CMTFPLS_LIB="$OUT/libcmtfpls_$name.so" python "$ROOT/tools/kernel_bench.py" --only mfma "${@:2}" 2>&1 | grep -v amdgpu.ids
  done
fi
