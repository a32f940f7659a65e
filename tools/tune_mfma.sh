#!/bin/bash
# A/B variants of the two f64-MFMA kernels (xcov, mttkrp): prefetch depth and grid size.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
VARIANTS=("m_base:" "m_un2:-DCMTFPLS_XCOV_UN=2 -DCMTFPLS_MTTKRP_UN=2" "m_un8:-DCMTFPLS_XCOV_UN=8 -DCMTFPLS_MTTKRP_UN=8"
          "m_b2048:-DCMTFPLS_XCOV_BLOCKS=2048" "m_b768:-DCMTFPLS_XCOV_BLOCKS=768" "m_b512:-DCMTFPLS_XCOV_BLOCKS=512")
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"; rm -f "$OUT"/libcmtfpls_m_*.so
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags "$SRC"/{runtime,sweeps,small,rank1,rank1_tensor,xcov,mttkrp,mixed}.hip -o "$OUT/libcmtfpls_$name.so" ) &
  done
  wait; ls "$OUT" | grep m_
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    echo "=== variant $name"
    CMTFPLS_LIB="$OUT/libcmtfpls_$name.so" python "$ROOT/tools/kernel_bench.py" --only mfma "${@:2}" 2>&1 | grep -v amdgpu.ids
  done
fi
