#!/bin/bash
# Build variant libraries of the sweep kernels (compile-time knobs) into cmtf_pls_amd/lib/variants/
# Usage: tools/tune_sweeps.sh build   (here, cross-compile)   |   tools/tune_sweeps.sh run (GPU box)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("base:" "f1024:-DCMTFPLS_CONTRACT_BLOCKS_FULL=1024" "f2048:-DCMTFPLS_CONTRACT_BLOCKS_FULL=2048" "f4096:-DCMTFPLS_CONTRACT_BLOCKS_FULL=4096" "f2048r4:-DCMTFPLS_CONTRACT_BLOCKS_FULL=2048 -DCMTFPLS_UNROLL_FULL=4")
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
        "$SRC"/*.hip -o "$OUT/libcmtfpls_$name.so" ) &
  done
  wait
  ls -la "$OUT"
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    echo "=== variant $name"
    CMTFPLS_LIB="$OUT/libcmtfpls_$name.so" python "$ROOT/tools/kernel_bench.py" --only "${ONLY:-sweeps}" "${@:2}" 2>&1 | grep -v amdgpu.ids
  done
fi
