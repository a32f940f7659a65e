// Mixed-precision variants of the two matrix-core contractions for f32-stored X (opt-in):
//   xcov_mixed    S = Y^T X_(0)            (see xcov.hip)
//   mttkrp_mixed  M = X_(0) (WA (.) WB)    (see mttkrp.hip)
// on v_mfma_f32_16x16x4_f32, which issues in half the cycles of v_mfma_f64_16x16x4_f64: the f64
// kernels are bound by the f64 matrix pipe (72-80 % busy at the clock the chip holds,
// profiles/r01i_mfma_utilisation.json), these are bound by HBM.
// Precision contract (why this is opt-in and the f64 kernels are the default): X is exact (it IS
// f32); the other operand (Y, or the Khatri-Rao entry wA*wB formed in f64) is rounded once to f32
// (relative 6e-8); products are accumulated in f32 only inside a chain of 64 rows (xcov) / 256 columns
// (mttkrp) and every chain is then added into f64 accumulators.  The f32 MFMA is a k-ordered fmaf
// chain (MI355X guide), so the result is deterministic.  Measured effect on a fit: see
// tests/test_gpu_mixed.py (scores within 2e-6 relative of the f64 path).
//
// Tile mapping is that of the f64 kernels except for the D layout of the f32 16x16 MFMA:
//   D[reg g] of lane l is row (l >> 4) * 4 + g, column l & 15   (f64: row (l >> 4) + 4 * g).
#include "common.hpp"

namespace cmtfpls {

typedef float f4_t __attribute__((ext_vector_type(4)));

#ifndef CMTFPLS_MIXED_UN
#define CMTFPLS_MIXED_UN 4
#endif
#ifndef CMTFPLS_MIXED_FAST
#define CMTFPLS_MIXED_FAST 1
#endif
#ifndef CMTFPLS_MIXED_UN_WIDE
#define CMTFPLS_MIXED_UN_WIDE 4
#endif

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st);

// FAST: every tile is interior (P % 256 == 0, M % 16 == 0, row blocks a multiple of 8 * UN rows): no
// clamps and no selects around the MFMAs.
template <bool MASKED, bool VEC, int MT, bool FAST>
__global__ __launch_bounds__(256) void xcov_mixed_kernel(const float* __restrict__ X, int64_t I, int64_t P,
                                                        const double* __restrict__ Y, int ldy, int M,
                                                        double* __restrict__ part, int rows_per_block) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, nn = lane & 15;
  const int64_t cb = ((int64_t)blockIdx.x * 4 + wv) * 64;
  if (cb >= P) return;
  const int64_t c = cb + 4 * nn;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < I) ? r0 + rows_per_block : I;
  using XV = Pack<float, 4>;
  f4_t acc32[MT][4];
  double acc64[MT][4][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc32[mt][e] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) acc64[mt][e][g] = 0.0;
    }
  bool mok[MT];
  int ycol[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { mok[mt] = (mt * 16 + nn) < M; ycol[mt] = mok[mt] ? mt * 16 + nn : M - 1; }
  // prefetch depth: two register stages of UN loads each (fewer at M > 16, where the accumulators
  // already take 96 registers and occupancy is what hides the HBM latency)
  constexpr int UN = (MT >= 2) ? CMTFPLS_MIXED_UN_WIDE : CMTFPLS_MIXED_UN;
  constexpr int FLUSH_TRIPS = 64 / (8 * UN);     // f32 chains of 64 rows
  const int64_t cc = (c < P) ? c : (VEC ? P - 4 : P - 1);

  auto load_stage = [&](XV (&x)[UN], double (&a)[UN][MT], int64_t r) {
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int64_t row = r + 4 * s + kq;
      const int64_t rowc = (FAST || row < r1) ? row : r1 - 1;
      if (VEC) {
        x[s] = ld_stream(reinterpret_cast<const XV*>(X + rowc * P + cc));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[s].e[e] = X[rowc * P + ((cc + e < P) ? cc + e : P - 1)];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[s][mt] = Y[rowc * ldy + ycol[mt]];   // rounded to f32 at use, not here:
                                                                             // a convert here would wait on the load
    }
  };
  auto mma_stage = [&](const XV (&x)[UN], const double (&a)[UN][MT], int64_t r) {
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const bool rok = FAST || (r + 4 * s + kq) < r1;
      float af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = (FAST || (rok && mok[mt])) ? (float)a[s][mt] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xv = x[s].e[e];
        if (MASKED) xv = (xv == xv) ? xv : 0.f;
        const float b = (FAST || (rok && c + e < P)) ? xv : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc32[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], b, acc32[mt][e], 0, 0, 0);
      }
    }
  };
  auto flush = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc64[mt][e][g] += (double)acc32[mt][e][g];
        acc32[mt][e] = f4_t{0.f, 0.f, 0.f, 0.f};
      }
  };

  XV xa[UN], xb[UN];
  double aa[UN][MT], ab[UN][MT];
  load_stage(xa, aa, r0);
  int trip = 0;
  for (int64_t r = r0; r < r1; r += 8 * UN, ++trip) {
    // the barriers pin "issue the whole next stage, then multiply the current one": left alone, the
    // scheduler sinks the loads next to their first use (vmcnt(0) straight after the load) when
    // registers are tight (M > 16), which serialises HBM latency with the matrix pipe
    load_stage(xb, ab, r + 4 * UN);
    __builtin_amdgcn_sched_barrier(0);
    mma_stage(xa, aa, r);
    __builtin_amdgcn_sched_barrier(0);
    // FAST has no clamp: the look-ahead of the last trip must stay inside this block's rows
    load_stage(xa, aa, (FAST && r + 8 * UN >= r1) ? r : r + 8 * UN);
    __builtin_amdgcn_sched_barrier(0);
    mma_stage(xb, ab, r + 4 * UN);
    __builtin_amdgcn_sched_barrier(0);
    if (trip % FLUSH_TRIPS == FLUSH_TRIPS - 1) flush();
  }
  flush();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = mt * 16 + kq * 4 + g;
      if (m < M) {
        double* dst = part + ((int64_t)blockIdx.y * M + m) * P + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < P) dst[e] = acc64[mt][e][g];
      }
    }
}

template <bool VEC, int RT>
__global__ __launch_bounds__(256) void mttkrp_mixed_kernel(const float* __restrict__ X, int64_t I, int A, int B,
                                                          const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                          double* __restrict__ out, int ldo) {
  extern __shared__ double lds[];
  constexpr int RP = 16 * RT;
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ri = lane & 15, kq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int64_t ngroups = (I + 15) / 16;
  using XV = Pack<float, 4>;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t i0 = grp * 16;
    const bool rok = (i0 + ri) < I;
    const float* __restrict__ xr = X + (rok ? i0 + ri : I - 1) * P;
    f4_t acc32[RT];
    double acc64[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      acc32[t] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) acc64[t][g] = 0.0;
    }
    KronWalk w(4 * kq, 16, B);
    constexpr int UN = 4;
    int trip = 0;
    for (int64_t c0 = 0; c0 < P; c0 += 16 * UN, ++trip) {
      XV x[UN];
#pragma unroll
      for (int s = 0; s < UN; ++s) {
        const int64_t c = c0 + 16 * s + 4 * kq;
        if (VEC) {
          x[s] = ld_stream(reinterpret_cast<const XV*>(xr + ((c < P) ? c : P - 4)));
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
          const bool ok = rok && (c + e < P);
          const float a = ok ? x[s].e[e] : 0.f;
          int j = w.j, k = w.k + e;
          if (!VEC && k >= B) { j += k / B; k = k % B; }
          const bool wok = (c + e < P);
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const float b = wok ? (float)(sA[(size_t)j * RP + t * 16 + ri] * sB[(size_t)k * RP + t * 16 + ri]) : 0.f;
            acc32[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc32[t], 0, 0, 0);
          }
        }
        w.next();
      }
      if ((trip & 3) == 3) {               // f32 chains of 256 columns
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
          for (int g = 0; g < 4; ++g) acc64[t][g] += (double)acc32[t][g];
          acc32[t] = f4_t{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t row = i0 + kq * 4 + g;
        const int r = t * 16 + ri;
        if (row < I && r < R) out[row * ldo + r] = acc64[t][g] + (double)acc32[t][g];
      }
  }
}

// The k-row form of mttkrp.hip (round 3: the matrix cores contract over the j-rows with the plain loading WA[j, r] as B operand,
// fully coalesced loads, WB folded into the accumulators afterwards) on the f32 matrix cores: X exact, WA[j, r] rounded once to
// f32, f32 accumulation inside a chain of at most 64 j-steps (256 terms), every chain multiplied by the f64 WB[k, r] into one f64
// double per lane.  Shapes as mttkrp_kj_kernel<float, NL>: B % (NL * 64) == 0, A % (32 / NL) == 0, R <= 16.
template <int NL>
__global__ __launch_bounds__(256) void mttkrp_kj_mixed_kernel(const float* __restrict__ X, int64_t I, int A, int B,
                                                              const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                              double* __restrict__ out, int ldo) {
  constexpr int V = 4, RP = 16, CHJ = 8 / NL, KP = NL * 16 * V;
  constexpr int FLUSH = 64 / CHJ;          // chunks per f32 chain (64 j-steps = 256 terms)
  extern __shared__ double lds[];          // sB[B][16] f64, then sA[A][16] as f32
  double* sB = lds;
  float* sA = reinterpret_cast<float*>(lds + (size_t)B * RP);
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? (float)WA[(int64_t)j * R + r] : 0.f; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kk = lane & 15, jq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int npass = B / KP, cpp = A / (4 * CHJ);
  const int nchunks = npass * cpp, npairs = nchunks / 2;
  const int64_t istep = (int64_t)gridDim.x * 4;
  using XV = Pack<float, V>;
  XV b0[CHJ][NL], b1[CHJ][NL];
  auto load = [&](XV (&buf)[CHJ][NL], const float* __restrict__ xs, int q) {
    const int ps = q / cpp, c = q - ps * cpp;
    const float* __restrict__ xp = xs + (int64_t)(c * CHJ * 4) * B + ps * KP;
#pragma unroll
    for (int h = 0; h < CHJ; ++h)
#pragma unroll
      for (int n = 0; n < NL; ++n) buf[h][n] = ld_stream(reinterpret_cast<const XV*>(xp + (int64_t)(h * 4) * B + n * 16 * V));
  };
  int64_t i = (int64_t)blockIdx.x * 4 + wv;
  const int64_t lane_off = (int64_t)jq * B + V * kk;
  if (i < I) load(b0, X + i * P + lane_off, 0);
  for (; i < I; i += istep) {
    const float* __restrict__ xs = X + i * P + lane_off;
    const float* __restrict__ xs_next = X + ((i + istep < I) ? i + istep : i) * P + lane_off;
    double s = 0.0;
    f4_t acc[NL][V];
#pragma unroll
    for (int n = 0; n < NL; ++n)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[n][e] = f4_t{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](XV (&buf)[CHJ][NL], int q) {
      const int ps = q / cpp, c = q - ps * cpp;
      const float* __restrict__ sa = sA + (size_t)(c * CHJ * 4 + jq) * RP + kk;
#pragma unroll
      for (int h = 0; h < CHJ; ++h) {
        const float wa = sa[(size_t)(h * 4) * RP];
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[n][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(buf[h][n].e[e], wa, acc[n][e], 0, 0, 0);
      }
      if (c == cpp - 1 || (c % FLUSH) == FLUSH - 1) {     // end of an f32 chain / of the pass: fold WB in (f64), fresh accumulators
        const double* __restrict__ sb = sB + (size_t)(ps * KP) * RP + kk;
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e) {
#pragma unroll
            for (int g = 0; g < 4; ++g) s = fma(sb[(size_t)(n * 16 * V + V * (4 * jq + g) + e) * RP], (double)acc[n][e][g], s);   // f32 D layout: row 4 (l >> 4) + g
            acc[n][e] = f4_t{0.f, 0.f, 0.f, 0.f};
          }
      }
    };
    for (int t = 0; t < npairs; ++t) {
      const int q = 2 * t;
      load(b1, xs, q + 1);
      compute(b0, q);
      const bool more = (q + 2 < nchunks);
      load(b0, more ? xs : xs_next, more ? q + 2 : 0);
      compute(b1, q + 1);
    }
    if (nchunks & 1) {
      compute(b0, nchunks - 1);
      load(b0, xs_next, 0);
    }
    s += __shfl_xor(s, 16, kWave);
    s += __shfl_xor(s, 32, kWave);
    if (jq == 0 && kk < R) out[i * ldo + kk] = s;
  }
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

int cmtfpls_xcov_f32_mixed(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, int masked,
                           void* ws, size_t ws_bytes, void* stream) {
  if (!X || !Y || !S || I <= 0 || P <= 0 || M <= 0 || ldy < M) { set_error("xcov_mixed: bad argument"); return CMTFPLS_EINVAL; }
  if (M > kXcovMaxResponses) {                            // tiles of <= 64 responses, one pass over X each (as cmtfpls_xcov_*)
    for (int lo = 0; lo < M; lo += kXcovMaxResponses) {
      const int mt = (M - lo < kXcovMaxResponses) ? M - lo : kXcovMaxResponses;
      const int rc = cmtfpls_xcov_f32_mixed(X, I, P, Y + lo, ldy, mt, S + (int64_t)lo * P, masked, ws, ws_bytes, stream);
      if (rc != CMTFPLS_OK) return rc;
    }
    return CMTFPLS_OK;
  }
  const XcovPlan p = plan_xcov(I, P);
  const size_t need = (size_t)p.row_blocks * M * P * sizeof(double);
  if (!ws || ws_bytes < need) { set_error("xcov_mixed: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const int mt = (M + 15) / 16;
  double* part = static_cast<double*>(ws);
  const dim3 grid(p.col_tiles, p.row_blocks), block(256);
  constexpr int kMaxUn = (CMTFPLS_MIXED_UN > CMTFPLS_MIXED_UN_WIDE) ? CMTFPLS_MIXED_UN : CMTFPLS_MIXED_UN_WIDE;
  const bool fast = CMTFPLS_MIXED_FAST && vec && (P % 256 == 0) && (M % 16 == 0) && (M / 16 != 3) && (I % p.rows_per_block == 0) &&
                    (p.rows_per_block % (8 * kMaxUn) == 0);
#define XL(MSK, VC, MTT, FS) hipLaunchKernelGGL((xcov_mixed_kernel<MSK, VC, MTT, FS>), grid, block, 0, st, X, I, P, Y, ldy, M, part, p.rows_per_block)
#define XM(MSK, VC, FS) do { if (mt == 1) XL(MSK, VC, 1, FS); else if (mt == 2) XL(MSK, VC, 2, FS); else XL(MSK, VC, 4, FS); } while (0)
  if (masked) { if (fast) XM(true, true, true); else if (vec) XM(true, true, false); else XM(true, false, false); }
  else        { if (fast) XM(false, true, true); else if (vec) XM(false, true, false); else XM(false, false, false); }
#undef XM
#undef XL
  launch_reduce_rows(part, p.row_blocks, (int64_t)M * P, S, st);
  return check_launch("xcov_mixed");
}

int cmtfpls_mttkrp_f32_mixed(const float* X, int64_t I, int A, int B, const double* WA, const double* WB, int R,
                             double* out, int ldo, void* stream) {
  if (!X || !WA || !WB || !out || I <= 0 || A <= 0 || B <= 0 || R <= 0 || ldo < R) { set_error("mttkrp_mixed: bad argument"); return CMTFPLS_EINVAL; }
  if (R > 32) { set_error("mttkrp_mixed: more than 32 components per call"); return CMTFPLS_EUNSUPPORTED; }
  const int rt = (R + 15) / 16;
  const size_t lds = (size_t)(A + B) * 16 * rt * sizeof(double);
  if (lds > 96 * 1024) { set_error("mttkrp_mixed: loadings exceed LDS"); return CMTFPLS_EUNSUPPORTED; }
  const bool vec = (B % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (R <= 16 && vec) {
    // the k-row form (coalesced loads, WA as the plain B operand): 128-column passes, or 64
    const size_t lds_kj = (size_t)B * 16 * sizeof(double) + (size_t)A * 16 * sizeof(float);
    int gk = (int)((I + 3) / 4);
    if (gk > 2048) gk = 2048;
    if (lds_kj <= 64 * 1024) {
      if (B % 128 == 0 && A % 16 == 0) {
        hipLaunchKernelGGL((mttkrp_kj_mixed_kernel<2>), dim3(gk), dim3(256), lds_kj, st, X, I, A, B, WA, WB, R, out, ldo);
        return check_launch("mttkrp_mixed");
      }
      if (B % 64 == 0 && A % 32 == 0) {
        hipLaunchKernelGGL((mttkrp_kj_mixed_kernel<1>), dim3(gk), dim3(256), lds_kj, st, X, I, A, B, WA, WB, R, out, ldo);
        return check_launch("mttkrp_mixed");
      }
    }
  }
  const int64_t ngroups = (I + 15) / 16;
  int grid = (int)((ngroups + 3) / 4);
  if (grid > 2048) grid = 2048;
#define ML(VC, RTT) hipLaunchKernelGGL((mttkrp_mixed_kernel<VC, RTT>), dim3(grid), dim3(256), lds, st, X, I, A, B, WA, WB, R, out, ldo)
  if (vec) { if (rt == 1) ML(true, 1); else ML(true, 2); }
  else     { if (rt == 1) ML(false, 1); else ML(false, 2); }
#undef ML
  return check_launch("mttkrp_mixed");
}

}  // extern "C"
