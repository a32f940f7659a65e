#!/bin/bash
# Build variant libraries of the sweep kernels (compile-time knobs) into cmtf_pls_amd/lib/variants/
# Usage: tools/tune_sweeps.sh build   (here, cross-compile)   |   tools/tune_sweeps.sh run (GPU box)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("base:" "k4:-DCMTFPLS_YQ_ROWS_IN_FLIGHT=4" "k8:-DCMTFPLS_YQ_ROWS_IN_FLIGHT=8" "cb1280:-DCMTFPLS_CONTRACT_BLOCKS=1280" "cb1280k4:-DCMTFPLS_CONTRACT_BLOCKS=1280 -DCMTFPLS_YQ_ROWS_IN_FLIGHT=4")
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
