// MTTKRP  M[i, r] = sum_c X[i, c] * WA[c / B, r] * WB[c % B, r]   (M = X_(0) (WA (.) WB), I x R, f64)
// on the f64 matrix cores, the Khatri-Rao operand formed on the fly from the two loading matrices
// staged in LDS (it is never materialised: P x R would be 1.3 MB at 128 x 128 x 10).
//
// Why: transform / predict (tpls.py:128-142, 151-165) project-and-deflate R times over a copy of X
// (R reads + R writes).  Without missing values the deflations are linear, X_{a+1} = X_a - t_a w_a^T,
// so  t_a = (X_0 w_a) - sum_{j<a} t_j (w_j . w_a)  and all scores follow from ONE pass over X:
// T = M (I + triu(W^T W, 1))^{-1}  (engine.project does the R x R part); with coupled blocks M and
// W^T W are averaged over blocks first (cmtf.py:155,206).
//
// Tile mapping (one MFMA = 16 rows of X x 16 components x 4 columns of X):
//   lane l: ri = l & 15 (row inside the 16-row group), kq = l >> 4.
//   the lane loads X[i0 + ri][c0 + 4*kq .. +3] as one 16-byte vector and feeds element e to MFMA e:
//   A operand of MFMA e = X[i0 + ri][c0 + 4*kq + e]                  (A[i = l&15][k = l>>4])
//   B operand of MFMA e = W[c0 + 4*kq + e][r = l & 15] = sA[j][r] * sB[k][r]   (B[k = l>>4][j = l&15])
//   all four MFMAs accumulate into the same D: D[(l>>4) + 4*g][l & 15] = M[i0 + (l>>4) + 4*g][r].
// One wavefront owns 16 whole rows (no partials, fixed summation order).
//
// Round-2 negative results (kept out of the code): (1) the operand layout makes every load instruction touch
// 16 rows x 64 B, a pattern that by itself tops out at 6.0 TB/s against 7.0-7.2 TB/s for row-contiguous reads
// (tools/exp/rowexp.hip kinds 40-43, profiles/r02i_mfma_access.txt); a variant that fetched 16 x 128-column tiles
// row-contiguously and transposed them through a wavefront-private LDS tile ran at the SAME 0.85-0.87 ms;
// (2) so did four independent accumulator chains instead of one (profiles/r02k_mttkrp_variants.txt); (3) with the
// Khatri-Rao operand replaced by a constant AND the f32 -> f64 conversion removed -- loads and MFMAs only -- the
// kernel still takes 0.77 ms (tools/exp/mttkrp_exp.hip, profiles/r02n_mttkrp_what_bounds_it.txt): 1.68e7 MFMAs on the
// 16-wide tile that R = 10 components occupy cost ~47 ns each per SIMD at the clock the chip holds under f64 MFMA
// load.  The LDS weight path adds 0.1 ms on top; sharing one B operand between two row groups recovers half of
// that (0.82 ms) and was not worth a second kernel form.
#include "common.hpp"

namespace cmtfpls {

typedef double d4m_t __attribute__((ext_vector_type(4)));

#ifndef CMTFPLS_MTTKRP_UN
#define CMTFPLS_MTTKRP_UN 4
#endif

// The loadings of both modes, padded to 16 components, must fit one workgroup's LDS: (A + B) * 16 * ceil(R / 16) doubles.
// Up to 64 KB two or more workgroups share a CU; beyond that one workgroup per CU (152 KB: A + B <= 1216 at R <= 16), which
// still beats the R read+write passes of the sequential path by a wide margin (profiles/r02an_mttkrp_big_lds.txt).
constexpr size_t kMttkrpLdsMax = 152 * 1024;

// FAST: I % 16 == 0 and P % (16 * UN) == 0 with vector loads: no clamps, no selects.
// NT threads per workgroup: the loadings are staged once per workgroup, so a workgroup whose LDS share allows only one or
// two of its kind per CU brings 8 or 16 wavefronts instead of 4 (every wavefront works on its own 16 rows: the result
// does not depend on NT).
template <typename T, bool VEC, int RT, bool FAST, int NT>
__global__ __launch_bounds__(NT) void mttkrp_kernel(const T* __restrict__ X, int64_t I, int A, int B,
                                                    const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                    double* __restrict__ out, int ldo) {
  extern __shared__ double lds[];          // sA[A][16*RT] then sB[B][16*RT], zero padded beyond R
  constexpr int RP = 16 * RT;
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  constexpr int NW = NT / 64;
  for (int idx = threadIdx.x; idx < A * RP; idx += NT) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += NT) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ri = lane & 15, kq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int64_t ngroups = (I + 15) / 16;
  using XV = Pack<T, 4>;
  for (int64_t grp = (int64_t)blockIdx.x * NW + wv; grp < ngroups; grp += (int64_t)gridDim.x * NW) {
    const int64_t i0 = grp * 16;
    const bool rok = FAST || (i0 + ri) < I;
    const T* __restrict__ xr = X + ((rok ? i0 + ri : I - 1)) * P;
    d4m_t acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = d4m_t{0.0, 0.0, 0.0, 0.0};
    KronWalk w(4 * kq, 16, B);
    constexpr int UN = CMTFPLS_MTTKRP_UN;
    for (int64_t c0 = 0; c0 < P; c0 += 16 * UN) {
      XV x[UN];
#pragma unroll
      for (int s = 0; s < UN; ++s) {
        const int64_t c = c0 + 16 * s + 4 * kq;
        if (VEC) {
          x[s] = ld_stream(reinterpret_cast<const XV*>(xr + ((FAST || c < P) ? c : P - 4)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[s].e[e] = xr[(c + e < P) ? c + e : P - 1];
        }
      }
#pragma unroll
      for (int s = 0; s < UN; ++s) {
        const int64_t c = c0 + 16 * s + 4 * kq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = FAST || (rok && (c + e < P));
          const double a = ok ? (double)x[s].e[e] : 0.0;
          int j = w.j, k = w.k + e;
          if (!VEC && k >= B) { j += k / B; k = k % B; }   // scalar path: a 4-column group may straddle a j boundary
          const bool wok = FAST || (c + e < P);
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const double b = wok ? sA[(size_t)j * RP + t * 16 + ri] * sB[(size_t)k * RP + t * 16 + ri] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
          }
        }
        w.next();
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t row = i0 + kq + 4 * g;
        const int r = t * 16 + ri;
        if (row < I && r < R) out[row * ldo + r] = acc[t][g];
      }
  }
}


template <typename T>
static int run_mttkrp(const T* X, int64_t I, int A, int B, const double* WA, const double* WB, int R, double* out, int ldo,
                      hipStream_t st) {
  if (!X || !WA || !WB || !out || I <= 0 || A <= 0 || B <= 0 || R <= 0 || ldo < R) { set_error("mttkrp: bad argument"); return CMTFPLS_EINVAL; }
  if (R > 32) { set_error("mttkrp: more than 32 components per call"); return CMTFPLS_EUNSUPPORTED; }
  const int rt = (R + 15) / 16;
  const size_t lds = (size_t)(A + B) * 16 * rt * sizeof(double);
  if (lds > kMttkrpLdsMax) { set_error("mttkrp: loadings exceed LDS"); return CMTFPLS_EUNSUPPORTED; }
  const bool vec = (B % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & (4 * sizeof(T) - 1)) == 0);
  const int64_t ngroups = (I + 15) / 16;
  // Threads per workgroup (profiles/r02ap_mttkrp_nt.txt): the LDS share decides how many workgroups a CU holds; take the
  // smallest workgroup that still puts 16 wavefronts on a CU (64 KB of loadings at 256 x 256: 512 threads, 4.8 -> 5.4 TB/s;
  // 96 KB: 3.1 -> 4.8 TB/s), but never so large that fewer than 256 workgroups are left for the 256 CUs (few, long rows).
#ifdef CMTFPLS_MTTKRP_NT
  const int nt = vec ? CMTFPLS_MTTKRP_NT : 256;                  // tuning builds only
#else
  int nt = 256;
  if (vec) {
    const int wgs_per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
    while (nt < 1024 && wgs_per_cu * (nt / 64) < 16) nt *= 2;
    while (nt > 256 && (ngroups + nt / 64 - 1) / (nt / 64) < 256) nt /= 2;
  }
#endif
  const int nw = nt / 64;
  int grid = (int)((ngroups + nw - 1) / nw);
  if (grid > 8192 / nw) grid = 8192 / nw;
  const dim3 g(grid), b(nt);
  const bool fast = vec && (I % 16 == 0) && (((int64_t)A * B) % (16 * CMTFPLS_MTTKRP_UN) == 0);
#define ML(VC, RTT, FS, NTT)                                                                                                   \
  do {                                                                                                                         \
    if (lds > 64 * 1024)                                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mttkrp_kernel<T, VC, RTT, FS, NTT>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                         \
    hipLaunchKernelGGL((mttkrp_kernel<T, VC, RTT, FS, NTT>), g, b, lds, st, X, I, A, B, WA, WB, R, out, ldo);                  \
  } while (0)
#define MLN(RTT, FS)                                                                                                           \
  do {                                                                                                                         \
    if (nt == 256) ML(true, RTT, FS, 256); else if (nt == 512) ML(true, RTT, FS, 512); else ML(true, RTT, FS, 1024);           \
  } while (0)
  if (fast) { if (rt == 1) MLN(1, true); else MLN(2, true); }
  else if (vec) { if (rt == 1) MLN(1, false); else MLN(2, false); }
  else     { if (rt == 1) ML(false, 1, false, 256); else ML(false, 2, false, 256); }
#undef MLN
#undef ML
  return check_launch("mttkrp");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
int cmtfpls_mttkrp_f32(const float* X, int64_t I, int A, int B, const double* WA, const double* WB, int R, double* out, int ldo, void* stream) {
  return run_mttkrp<float>(X, I, A, B, WA, WB, R, out, ldo, (hipStream_t)stream);
}
int cmtfpls_mttkrp_f64(const double* X, int64_t I, int A, int B, const double* WA, const double* WB, int R, double* out, int ldo, void* stream) {
  return run_mttkrp<double>(X, I, A, B, WA, WB, R, out, ldo, (hipStream_t)stream);
}
}
