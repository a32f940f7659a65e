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
for src in runtime sweeps small rank1 rank1_tensor xcov mttkrp mixed ceiling solve recon synth loo loo_xcov collective project scorecontract; do
  "$HIPCC" "${FLAGS[@]}" -c "$HERE/$src.hip" -o "$OUT/$src.o" &
  pids+=($!)
  objs+=("$OUT/$src.o")
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcmtfpls.so" "${objs[@]}" -ldl
echo "built $OUT/libcmtfpls.so"
# torch-free demo of the C ABI (run by tests/test_gpu_c_abi_demo.py on the GPU box)
ROOT="$HERE/../.."
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 "$ROOT/examples/c_abi_demo.cpp" -L"$OUT" -lcmtfpls -ldl \
  -Wl,-rpath,'$ORIGIN/../cmtf_pls_amd/lib' -o "$ROOT/examples/c_abi_demo"
echo "built $ROOT/examples/c_abi_demo"
