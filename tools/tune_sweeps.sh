#!/bin/bash
# Build variant libraries of the sweep kernels (compile-time knobs) into cmtf_pls_amd/lib/variants/
# Usage: tools/tune_sweeps.sh build   (here, cross-compile)   |   tools/tune_sweeps.sh run (GPU box)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("base:" "b256:-DCMTFPLS_SWEEP_BLOCKS=256" "b768:-DCMTFPLS_SWEEP_BLOCKS=768" "b2048:-DCMTFPLS_SWEEP_BLOCKS=2048" "cb2048:-DCMTFPLS_CONTRACT_BLOCKS=2048" "ru4:-DCMTFPLS_ROW_UNROLL=4"
          "nopad:-DCMTFPLS_DEFLATE_ROWS_PAD=0" "norows:-DCMTFPLS_DEFLATE_ROWS=0" "ntoff:-DCMTFPLS_NT_LOAD=0 -DCMTFPLS_NT_STORE=0")
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
