#!/bin/bash
# The 8192-row shard a GPU sees at N = 8 (VERDICT r3 "Next" #4): compile-time variants of the two read sweeps, timed inside
# bench.py at --shape 8192 128 128 (graph replay, the product's iteration).
# Usage: tools/tune_small_shard.sh build   (here, cross-compile)   |   tools/tune_small_shard.sh run [bench flags]   (GPU box)
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=(
  "sb1024:-DCMTFPLS_SWEEP_BLOCKS=1024"
  "sb256:-DCMTFPLS_SWEEP_BLOCKS=256"
  "ruf4:-DCMTFPLS_UNROLL_FULL=4"
  "cbf256:-DCMTFPLS_CONTRACT_BLOCKS_FULL=256"
  "cbf1024:-DCMTFPLS_CONTRACT_BLOCKS_FULL=1024"
  "cbf256ruf4:-DCMTFPLS_CONTRACT_BLOCKS_FULL=256 -DCMTFPLS_UNROLL_FULL=4"
  "ru16:-DCMTFPLS_ROW_UNROLL=16"
)
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags "$SRC"/*.hip -o "$OUT/libcmtfpls_$name.so" -ldl ) &
    while [ "$(jobs -r | wc -l)" -ge 3 ]; do sleep 1; done
  done
  wait
  ls -la "$OUT"
else
  names="base"
  for v in "${VARIANTS[@]}"; do names="$names ${v%%:*}"; done
  VARIANTS="$names" "$ROOT/tools/ab_bench.sh" --shape 8192 128 128 --steps 200 --warmup 20 "${@:2}"
fi
