#!/bin/bash
# A/B variants of the two f64-MFMA kernels (xcov, mttkrp): prefetch depth and grid size.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/cmtf_pls_amd/csrc"
OUT="$ROOT/cmtf_pls_amd/lib/variants"
if [ "${SET:-r2}" = forms ]; then   # round 3: the four MTTKRP forms
VARIANTS=("m_base:" "m_kj16:-DCMTFPLS_MTTKRP_NO_KJ4" "m_jk:-DCMTFPLS_MTTKRP_NO_KJ" "m_tile:-DCMTFPLS_MTTKRP_TILE_ONLY"
          "m_kj4r16:-DCMTFPLS_MTTKRP_KJ4_MAXR=16")     # the 4x4x4 form with FOUR component groups (same flops as the 16x16x4 tile)
elif [ "${SET:-r2}" = skip ]; then   # round 3 timing experiment: what does the j-block MTTKRP cost with 1/2, 1/4, 0 of its MFMAs?
# (the MFMA-skipping timing variants of round 3 -- CMTFPLS_MTTKRP_EXP_SKIP, wrong results by design -- were removed from the
#  kernel source in round 4; their measurements are in profiles/r03p_mttkrp_forms.txt)
VARIANTS=("m_base:" "m_jk:-DCMTFPLS_MTTKRP_NO_KJ" "m_tile:-DCMTFPLS_MTTKRP_TILE_ONLY")
elif [ "${SET:-r2}" = jk ]; then   # round 3: forms of the j-block MTTKRP
VARIANTS=("m_base:" "m_nobreg:-DCMTFPLS_MTTKRP_NOBREG" "m_acc2:-DCMTFPLS_MTTKRP_NACC=2" "m_acc4:-DCMTFPLS_MTTKRP_NACC=4"
          "m_acc4nobreg:-DCMTFPLS_MTTKRP_NACC=4 -DCMTFPLS_MTTKRP_NOBREG" "m_tile:-DCMTFPLS_MTTKRP_TILE_ONLY")
else
VARIANTS=("m_base:" "m_un2:-DCMTFPLS_XCOV_UN=2 -DCMTFPLS_MTTKRP_UN=2" "m_un8:-DCMTFPLS_XCOV_UN=8 -DCMTFPLS_MTTKRP_UN=8"
          "m_b2048:-DCMTFPLS_XCOV_BLOCKS=2048" "m_b768:-DCMTFPLS_XCOV_BLOCKS=768" "m_b512:-DCMTFPLS_XCOV_BLOCKS=512")
fi
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"; rm -f "$OUT"/libcmtfpls_m_*.so
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    # only the two matrix-core units are recompiled; every other object comes from the regular build (csrc/build.sh)
    ( for u in xcov mttkrp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$SRC/$u.hip" -o "$OUT/${name}_$u.o"; done
      others=$(ls "$ROOT"/cmtf_pls_amd/lib/*.o | grep -v "/xcov.o\|/mttkrp.o")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcmtfpls_$name.so" "$OUT/${name}_xcov.o" "$OUT/${name}_mttkrp.o" $others -ldl ) &
  done
  wait; ls "$OUT" | grep m_
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    echo "=== variant $name"
    CMTFPLS_LIB="$OUT/libcmtfpls_$name.so" python "$ROOT/tools/kernel_bench.py" --only mfma "${@:2}" 2>&1 | grep -v amdgpu.ids
  done
fi
