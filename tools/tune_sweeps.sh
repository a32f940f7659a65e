#!/bin/bash
# Build variant libraries of the sweep kernels (compile-time knobs) into cmtf_pls_amd/lib/variants/
# Usage: tools/tune_sweeps.sh build   (here, cross-compile)   |   tools/tune_sweeps.sh run (GPU box)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("base:" "u4:-DCMTFPLS_CONTRACT_U=4" "u4b512:-DCMTFPLS_CONTRACT_U=4 -DCMTFPLS_CONTRACT_BLOCKS=512" "u4r2:-DCMTFPLS_CONTRACT_U=4 -DCMTFPLS_UNROLL=2" "u1:-DCMTFPLS_CONTRACT_U=1" "u1r8:-DCMTFPLS_CONTRACT_U=1 -DCMTFPLS_UNROLL=8" "r8:-DCMTFPLS_UNROLL=8")
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
        "$SRC/runtime.hip" "$SRC/sweeps.hip" "$SRC/small.hip" "$SRC/rank1.hip" "$SRC/rank1_tensor.hip" "$SRC/xcov.hip" "$SRC/mttkrp.hip" "$SRC/mixed.hip" -o "$OUT/libcmtfpls_$name.so" ) &
  done
  wait
  ls -la "$OUT"
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    echo "=== variant $name"
    CMTFPLS_LIB="$OUT/libcmtfpls_$name.so" python "$ROOT/tools/kernel_bench.py" --only sweeps "${@:2}" 2>&1 | grep -v amdgpu.ids
  done
fi
