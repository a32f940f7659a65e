#!/bin/bash
# Build libcmtfpls.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function)
objs=()
pids=()
for src in runtime sweeps small rank1 rank1_tensor xcov mttkrp; do
  "$HIPCC" "${FLAGS[@]}" -c "$HERE/$src.hip" -o "$OUT/$src.o" &
  pids+=($!)
  objs+=("$OUT/$src.o")
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcmtfpls.so" "${objs[@]}"
echo "built $OUT/libcmtfpls.so"
