// Cross-covariance contraction  S[m, c] = sum_i Y[i, m] * X[i, c]   (S = X_(0)^T Y, M x P, f64)
// on the matrix cores: v_mfma_f64_16x16x4_f64, X converted on load, f64 accumulation throughout.
//
// Why it exists: inside one component X and Y are fixed and u = Y q, so the
// mode-0 contraction of every NIPALS iteration is Z = X x_0 u = sum_m q_m S_m and the Y update is
// Y^T t = S_(0) (w_J (x) w_K): the whole inner loop of tpls.py:79-107 can run on S (1-17 MB) and X is
// read once per component here instead of twice per iteration.  The result is the same arithmetic
// re-associated; engine.py offers it as algorithm="xcov".
//
// Tile mapping (64-wide wavefront, one MFMA = 16 (m) x 16 (cols) x 4 (rows of X)):
//   lane l: kq = l >> 4 (row of X inside the 4-row step), nn = l & 15.
//   A operand  = Y[r + kq][16*mt + nn]                       (A[i = l&15][k = l>>4])
//   the lane loads X[r + kq][cb + 4*nn .. 4*nn+3] as ONE 16-byte vector (a wave reads 4 rows x 256
//   contiguous bytes per instruction) and feeds element e to MFMA e, whose output column nn is
//   therefore X column cb + 4*nn + e:  B operand of MFMA e = X[r + kq][cb + 4*nn + e].
//   D tile e, register g: S[16*mt + kq + 4*g][cb + 4*nn + e]  (f64 map: col = l&15, row = (l>>4) + 4*g)
//   so each lane ends with 4 consecutive columns per (mt, g): 32-byte stores, 512 B per 16 lanes.
// Rows are split over gridDim.y row blocks -> (row block, M, P) f64 partials, summed in fixed order.
#include "common.hpp"

namespace cmtfpls {

typedef double d4_t __attribute__((ext_vector_type(4)));

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st);

XcovPlan plan_xcov(int64_t I, int64_t P) {
  XcovPlan p;
  p.col_tiles = (int)((P + 255) / 256);                 // 4 waves x 64 columns per workgroup
#ifndef CMTFPLS_XCOV_BLOCKS
#define CMTFPLS_XCOV_BLOCKS 1024
#endif
  int64_t want = (CMTFPLS_XCOV_BLOCKS + p.col_tiles - 1) / p.col_tiles;
  if (want < 1) want = 1;
  int64_t rpb = (I + want - 1) / want;
  rpb = (rpb + 63) / 64 * 64;                            // multiple of the 4-row step x unroll x 2 stages
  if (rpb < 64) rpb = 64;
  p.rows_per_block = (int)rpb;
  p.row_blocks = (int)((I + rpb - 1) / rpb);
  if (p.row_blocks < 1) p.row_blocks = 1;
  return p;
}

// FAST: every tile is interior (P % 256 == 0, M % 16 == 0, every row block a multiple of 32 rows):
// no clamps and no selects, so the only VALU work per MFMA is the f32 -> f64 conversion.
// SSQ (round 3): the same read of X also gives sum (x - mean[c])^2 -- |X - X_mean|^2, the denominator of R2X (tpls.py:115-117),
// for a fit that runs on the caller's UNCENTRED tensor (engine.FitRun.raw): the f64 value of every element is formed for the
// MFMA anyway, so it costs a subtraction and an fma per element and saves the separate pass (cmtfpls_recon_r2_* against a zero
// reconstruction).  One partial per wavefront, summed in index order by sum_kernel.
// STATS (round 4): the same read of X also gives every column's sum and sum of squares (one add and one fma per element on values
// the MFMA needs anyway) -- the statistics pass of tpls.py:61-71 for a block WITHOUT missing values (a NaN shows in its column's
// sum), so that a fit on the uncentred tensor reads X once before its first component instead of twice.  Per row block one
// partial row of 2 P doubles [sums | sums of squares] at ssq_part, summed in block order by reduce_rows_kernel.
template <typename T, bool MASKED, bool VEC, int MT, bool FAST, bool SSQ = false, bool STATS = false>
__global__ __launch_bounds__(256) void xcov_kernel(const T* __restrict__ X, int64_t I, int64_t P,
                                                  const double* __restrict__ Y, int ldy, int M,
                                                  double* __restrict__ part, int rows_per_block,
                                                  const double* __restrict__ mean = nullptr, double* __restrict__ ssq_part = nullptr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, nn = lane & 15;
  const int64_t cb = ((int64_t)blockIdx.x * 4 + wv) * 64;
  if (cb >= P) {                                         // whole wavefront past the last column
    if (SSQ && lane == 0) ssq_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv] = 0.0;
    return;
  }
  const int64_t c = cb + 4 * nn;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < I) ? r0 + rows_per_block : I;
  using XV = Pack<T, 4>;
  d4_t acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[mt][e] = d4_t{0.0, 0.0, 0.0, 0.0};
  bool mok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) mok[mt] = (mt * 16 + nn) < M;

  // Every load is unconditional (clamped address) and masked afterwards: a guarded load makes hipcc
  // branch around it and wait vmcnt(0) per load, which serialises the whole stream.  Two register
  // stages: the loads of the next 16-row step are in flight while the MFMAs of the current one run.
#ifndef CMTFPLS_XCOV_UN
#define CMTFPLS_XCOV_UN 4
#endif
  constexpr int UN = CMTFPLS_XCOV_UN;
  const int64_t cc = (c < P) ? c : (VEC ? P - 4 : P - 1);
  int ycol[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ycol[mt] = mok[mt] ? mt * 16 + nn : M - 1;
  double mu[SSQ ? 4 : 1], ssq = 0.0;
  double cs[STATS ? 4 : 1] = {}, cq[STATS ? 4 : 1] = {};
  if constexpr (SSQ) {
#pragma unroll
    for (int e = 0; e < 4; ++e) mu[e] = mean[(c + e < P) ? c + e : P - 1];
  }

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
      for (int mt = 0; mt < MT; ++mt) a[s][mt] = Y[rowc * ldy + ycol[mt]];
    }
  };
  auto mma_stage = [&](const XV (&x)[UN], const double (&a)[UN][MT], int64_t r) {
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const bool rok = FAST || (r + 4 * s + kq) < r1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T xv = x[s].e[e];
        if (MASKED) xv = (xv == xv) ? xv : (T)0;
        const double b = (FAST || (rok && c + e < P)) ? (double)xv : 0.0;
        if constexpr (SSQ) {
          const double dv = (FAST || (rok && c + e < P)) ? b - mu[e] : 0.0;
          ssq = fma(dv, dv, ssq);
        }
        if constexpr (STATS) {
          cs[e] += b;
          cq[e] = fma(b, b, cq[e]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][e] = __builtin_amdgcn_mfma_f64_16x16x4f64((FAST || (rok && mok[mt])) ? a[s][mt] : 0.0, b, acc[mt][e], 0, 0, 0);
      }
    }
  };

  XV xa[UN], xb[UN];
  double aa[UN][MT], ab[UN][MT];
  load_stage(xa, aa, r0);
  for (int64_t r = r0; r < r1; r += 8 * UN) {
    load_stage(xb, ab, r + 4 * UN);       // rows past r1 are clamped on load and masked in the MFMAs
    mma_stage(xa, aa, r);
    // FAST has no clamp: the look-ahead of the last trip must stay inside this block's rows
    load_stage(xa, aa, (FAST && r + 8 * UN >= r1) ? r : r + 8 * UN);
    mma_stage(xb, ab, r + 4 * UN);
  }
  if constexpr (SSQ) {
    ssq = wave_sum(ssq);
    if (lane == 0) ssq_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv] = ssq;
  }
  if constexpr (STATS) {                                 // the four lane groups hold the same columns for different rows
    double* srow = ssq_part + (int64_t)blockIdx.y * 2 * P;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double v = cs[e], w = cq[e];
      v += __shfl_xor(v, 16, kWave);
      v += __shfl_xor(v, 32, kWave);
      w += __shfl_xor(w, 16, kWave);
      w += __shfl_xor(w, 32, kWave);
      if (kq == 0 && c + e < P) { srow[c + e] = v; srow[P + c + e] = w; }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = mt * 16 + kq + 4 * g;
      if (m < M) {
        double* dst = part + ((int64_t)blockIdx.y * M + m) * P + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < P) dst[e] = acc[mt][e][g];
      }
    }
}

// Deflation AND the next component's cross-covariance in one read + write of X (round 3, blocks WITH missing values, whose S
// cannot be carried algebraically -- the masked deflation is not a rank-one update of S):
//   X[i,c] <- (T)(X[i,c] - (t[i] wA[c / B]) wB[c % B])      tpls.py:109; NaN (missing) stays NaN; the same fma as deflate_rows_kernel
//   S[m,c]  = sum_i Y[i,m] X0[i,c],  X0 = the deflated, ROUNDED X with NaN -> 0   (what cmtfpls_xcov_* would read afterwards)
//   ssq     = sum X0^2                                       the numerator of R2X (tpls.py:115-117)
// Same tiling, staging and MFMA order as xcov_kernel<T, true, true, MT, FAST>: S is bit-identical to deflating first and building
// S from the result.  A workgroup owns its (row block, 256 columns) region of X: no other workgroup reads or writes it.
template <typename T, int MT, bool FAST>
__global__ __launch_bounds__(256) void xcov_deflate_kernel(T* __restrict__ X, int64_t I, int64_t P, int B,
                                                          const double* __restrict__ Y, int ldy, int M,
                                                          double* __restrict__ part, int rows_per_block,
                                                          const double* __restrict__ t, const double* __restrict__ wA,
                                                          const double* __restrict__ wB, double* __restrict__ ssq_part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, nn = lane & 15;
  const int64_t cb = ((int64_t)blockIdx.x * 4 + wv) * 64;
  if (cb >= P) {                                         // whole wavefront past the last column
    if (lane == 0) ssq_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv] = 0.0;
    return;
  }
  const int64_t c = cb + 4 * nn;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < I) ? r0 + rows_per_block : I;
  using XV = Pack<T, 4>;
  d4_t acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[mt][e] = d4_t{0.0, 0.0, 0.0, 0.0};
  bool mok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) mok[mt] = (mt * 16 + nn) < M;
  constexpr int UN = CMTFPLS_XCOV_UN;
  const bool cok = c < P;                                // (P % 4 == 0: the lane's 4 columns exist together)
  const int64_t cc = cok ? c : P - 4;
  int ycol[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ycol[mt] = mok[mt] ? mt * 16 + nn : M - 1;
  double wa[4], wb[4], ssq = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    wa[e] = wA[(cc + e) / B];
    wb[e] = wB[(cc + e) % B];
  }

  auto load_stage = [&](XV (&x)[UN], double (&a)[UN][MT], double (&ts)[UN], int64_t r) {
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int64_t row = r + 4 * s + kq;
      const int64_t rowc = (FAST || row < r1) ? row : r1 - 1;
      x[s] = ld_stream(reinterpret_cast<const XV*>(X + rowc * P + cc));
      ts[s] = t[rowc];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[s][mt] = Y[rowc * ldy + ycol[mt]];
    }
  };
  auto mma_stage = [&](XV (&x)[UN], const double (&a)[UN][MT], const double (&ts)[UN], int64_t r) {
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int64_t row = r + 4 * s + kq;
      const bool rok = FAST || row < r1;                     // (the lane's Y values go with its ROW, whatever its columns)
      const bool ok = rok && (FAST || cok);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double tw = ts[s] * wa[e];
        const T nv = (T)fma(-tw, wb[e], (double)x[s].e[e]);
        x[s].e[e] = nv;
        const double b = (ok && nv == nv) ? (double)nv : 0.0;
        ssq = fma(b, b, ssq);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][e] = __builtin_amdgcn_mfma_f64_16x16x4f64((FAST || (rok && mok[mt])) ? a[s][mt] : 0.0, b, acc[mt][e], 0, 0, 0);
      }
      if (ok) st_stream(reinterpret_cast<XV*>(X + row * P + c), x[s]);
    }
  };

  XV xa[UN], xb[UN];
  double aa[UN][MT], ab[UN][MT], ta[UN], tb[UN];
  load_stage(xa, aa, ta, r0);
  for (int64_t r = r0; r < r1; r += 8 * UN) {
    load_stage(xb, ab, tb, r + 4 * UN);     // rows past r1 are clamped on load and masked (never stored) afterwards
    mma_stage(xa, aa, ta, r);
    // unconditional (a load behind a branch makes the compiler wait for every load); the last trip's look-ahead re-reads this
    // trip's own rows, already stored by this lane, and nothing uses what it returns
    load_stage(xa, aa, ta, (r + 8 * UN >= r1) ? r : r + 8 * UN);
    mma_stage(xb, ab, tb, r + 4 * UN);
  }
  ssq = wave_sum(ssq);
  if (lane == 0) ssq_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv] = ssq;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = mt * 16 + kq + 4 * g;
      if (m < M) {
        double* dst = part + ((int64_t)blockIdx.y * M + m) * P + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < P) dst[e] = acc[mt][e][g];
      }
    }
}

// du2 = (q - q_old)^T G (q - q_old):  |Y q - Y q_old|^2 with G = Y^T Y  (tpls.py:103 without
// touching the I-long vectors)
__global__ __launch_bounds__(256) void quadform_kernel(const double* __restrict__ G, int M, const double* __restrict__ q,
                                                      const double* __restrict__ q_old, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int idx = threadIdx.x; idx < M * M; idx += 256) {
    const int i = idx / M, j = idx - i * M;
    s = fma((q[i] - q_old[i]) * G[idx], (q[j] - q_old[j]), s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

static size_t xcov_ssq_extra(const XcovPlan& p) { return ((size_t)p.row_blocks * p.col_tiles * 4 * sizeof(double) + 255) / 256 * 256; }

template <typename T>
static int run_xcov_tile(const T* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, int masked,
                         void* ws, size_t ws_bytes, hipStream_t st, const double* mean = nullptr, double* ssq_out = nullptr,
                         double* stats_out = nullptr) {
  const XcovPlan p = plan_xcov(I, P);
  const bool with_ssq = ssq_out != nullptr, with_stats = stats_out != nullptr;
  if (with_ssq && (masked || !mean)) { set_error("xcov_ssq: needs the column means and a block without missing values"); return CMTFPLS_EINVAL; }
  if (with_stats && (masked || with_ssq)) { set_error("xcov_stats: a block without missing values, no second norm"); return CMTFPLS_EINVAL; }
  const size_t need = (size_t)p.row_blocks * M * P * sizeof(double) + (with_ssq ? xcov_ssq_extra(p) : 0) +
                      (with_stats ? (size_t)p.row_blocks * 2 * P * sizeof(double) : 0);
  if (!ws || ws_bytes < need) { set_error("xcov: workspace too small"); return CMTFPLS_EWORKSPACE; }
  const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & (4 * sizeof(T) - 1)) == 0);
  const int mt = (M + 15) / 16;                          // 1, 2, 3 -> 4, 4
  double* part = static_cast<double*>(ws);
  const dim3 grid(p.col_tiles, p.row_blocks), block(256);
  const bool fast = vec && (P % 256 == 0) && (M % 16 == 0) && (M / 16 != 3) && (I % p.rows_per_block == 0) &&
                    (p.rows_per_block % (8 * 4) == 0);
  double* ssq_part = (with_ssq || with_stats) ? reinterpret_cast<double*>(static_cast<char*>(ws) + (size_t)p.row_blocks * M * P * sizeof(double)) : nullptr;
#define XL(MSK, VC, MTT, FS) hipLaunchKernelGGL((xcov_kernel<T, MSK, VC, MTT, FS>), grid, block, 0, st, X, I, P, Y, ldy, M, part, p.rows_per_block, (const double*)nullptr, (double*)nullptr)
#define XS(VC, MTT, FS) hipLaunchKernelGGL((xcov_kernel<T, false, VC, MTT, FS, true>), grid, block, 0, st, X, I, P, Y, ldy, M, part, p.rows_per_block, mean, ssq_part)
#define XM(MSK, VC, FS) do { if (mt == 1) XL(MSK, VC, 1, FS); else if (mt == 2) XL(MSK, VC, 2, FS); else XL(MSK, VC, 4, FS); } while (0)
#define XQ(VC, FS) do { if (mt == 1) XS(VC, 1, FS); else if (mt == 2) XS(VC, 2, FS); else XS(VC, 4, FS); } while (0)
#define XT1(VC, MTT, FS) hipLaunchKernelGGL((xcov_kernel<T, false, VC, MTT, FS, false, true>), grid, block, 0, st, X, I, P, Y, ldy, M, part, p.rows_per_block, (const double*)nullptr, ssq_part)
#define XT(VC, FS) do { if (mt == 1) XT1(VC, 1, FS); else if (mt == 2) XT1(VC, 2, FS); else XT1(VC, 4, FS); } while (0)
  if (with_stats)  { if (fast) XT(true, true); else if (vec) XT(true, false); else XT(false, false); }
  else if (with_ssq) { if (fast) XQ(true, true); else if (vec) XQ(true, false); else XQ(false, false); }
  else if (masked) { if (fast) XM(true, true, true); else if (vec) XM(true, true, false); else XM(true, false, false); }
  else             { if (fast) XM(false, true, true); else if (vec) XM(false, true, false); else XM(false, false, false); }
#undef XT
#undef XT1
#undef XQ
#undef XM
#undef XS
#undef XL
  launch_reduce_rows(part, p.row_blocks, (int64_t)M * P, S, st);
  if (with_stats) launch_reduce_rows(ssq_part, p.row_blocks, 2 * P, stats_out, st);
  int rc = check_launch("xcov");
  if (rc == CMTFPLS_OK && with_ssq) rc = cmtfpls_sum_f64(ssq_part, (int64_t)p.row_blocks * p.col_tiles * 4, ssq_out, st);
  return rc;
}

// The kernel holds the accumulators of <= 64 responses (4 tiles of 16) per wavefront.  The reference has no limit on the number
// of responses (tpls.py:100-102): more of them are served in tiles of <= 64 columns of Y, one pass over X each, into the
// matching rows of S, through the same workspace (the passes are ordered on the stream); the norm comes out of the first pass.
template <typename T>
static int run_xcov(const T* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, int masked,
                    void* ws, size_t ws_bytes, hipStream_t st, const double* mean = nullptr, double* ssq_out = nullptr) {
  if (!X || !Y || !S || I <= 0 || P <= 0 || M <= 0 || ldy < M) { set_error("xcov: bad argument"); return CMTFPLS_EINVAL; }
  for (int lo = 0; lo < M; lo += kXcovMaxResponses) {
    const int mt = (M - lo < kXcovMaxResponses) ? M - lo : kXcovMaxResponses;
    const int rc = run_xcov_tile<T>(X, I, P, Y + lo, ldy, mt, S + (int64_t)lo * P, masked, ws, ws_bytes, st, mean, lo == 0 ? ssq_out : nullptr);
    if (rc != CMTFPLS_OK) return rc;
  }
  return CMTFPLS_OK;
}

template <typename T>
static int run_xcov_deflate(T* X, int64_t I, int A, int B, const double* Y, int ldy, int M, const double* t, const double* wA,
                            const double* wB, double* S, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!X || !Y || !S || !t || !wA || !wB || !ssq || I <= 0 || A <= 0 || B <= 0 || M <= 0 || ldy < M) {
    set_error("xcov_deflate: bad argument");
    return CMTFPLS_EINVAL;
  }
  const int64_t P = (int64_t)A * B;
  if (M > 64 || (P % 4) != 0 || (reinterpret_cast<uintptr_t>(X) & (4 * sizeof(T) - 1)) != 0) {
    set_error("xcov_deflate: M > 64 or rows that are not whole 4-element vectors; deflate, then xcov");
    return CMTFPLS_EUNSUPPORTED;
  }
  const XcovPlan p = plan_xcov(I, P);
  const size_t need = (size_t)p.row_blocks * M * P * sizeof(double) + xcov_ssq_extra(p);
  if (!ws || ws_bytes < need) { set_error("xcov_deflate: workspace too small"); return CMTFPLS_EWORKSPACE; }
  const int mt = (M + 15) / 16;
  double* part = static_cast<double*>(ws);
  double* ssq_part = reinterpret_cast<double*>(static_cast<char*>(ws) + (size_t)p.row_blocks * M * P * sizeof(double));
  const dim3 grid(p.col_tiles, p.row_blocks), block(256);
  const bool fast = (P % 256 == 0) && (M % 16 == 0) && (M / 16 != 3) && (I % p.rows_per_block == 0) && (p.rows_per_block % (8 * 4) == 0);
#define XD(MTT, FS) hipLaunchKernelGGL((xcov_deflate_kernel<T, MTT, FS>), grid, block, 0, st, X, I, P, B, Y, ldy, M, part, p.rows_per_block, t, wA, wB, ssq_part)
  if (fast) { if (mt == 1) XD(1, true); else if (mt == 2) XD(2, true); else XD(4, true); }
  else      { if (mt == 1) XD(1, false); else if (mt == 2) XD(2, false); else XD(4, false); }
#undef XD
  launch_reduce_rows(part, p.row_blocks, (int64_t)M * P, S, st);
  int rc = check_launch("xcov_deflate");
  if (rc == CMTFPLS_OK) rc = cmtfpls_sum_f64(ssq_part, (int64_t)p.row_blocks * p.col_tiles * 4, ssq, st);
  return rc;
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_xcov_workspace_bytes(int64_t I, int64_t P, int M) {
  if (I <= 0 || P <= 0 || M <= 0) return 0;
  const XcovPlan p = plan_xcov(I, P);
  if (M > kXcovMaxResponses) M = kXcovMaxResponses;       // (more responses: tiles of <= 64 through the same workspace)
  return (size_t)p.row_blocks * M * P * sizeof(double);
}
int cmtfpls_xcov_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, int masked,
                     void* ws, size_t ws_bytes, void* stream) {
  return run_xcov<float>(X, I, P, Y, ldy, M, S, masked, ws, ws_bytes, (hipStream_t)stream);
}
int cmtfpls_xcov_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, int masked,
                     void* ws, size_t ws_bytes, void* stream) {
  return run_xcov<double>(X, I, P, Y, ldy, M, S, masked, ws, ws_bytes, (hipStream_t)stream);
}
size_t cmtfpls_xcov_ssq_workspace_bytes(int64_t I, int64_t P, int M) {
  if (I <= 0 || P <= 0 || M <= 0) return 0;
  const XcovPlan p = plan_xcov(I, P);
  if (M > kXcovMaxResponses) M = kXcovMaxResponses;
  return (size_t)p.row_blocks * M * P * sizeof(double) + xcov_ssq_extra(p);
}
int cmtfpls_xcov_ssq_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, const double* mean,
                         double* ssq, void* ws, size_t ws_bytes, void* stream) {
  if (!mean || !ssq) { set_error("xcov_ssq: bad argument"); return CMTFPLS_EINVAL; }
  return run_xcov<float>(X, I, P, Y, ldy, M, S, 0, ws, ws_bytes, (hipStream_t)stream, mean, ssq);
}
int cmtfpls_xcov_ssq_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, const double* mean,
                         double* ssq, void* ws, size_t ws_bytes, void* stream) {
  if (!mean || !ssq) { set_error("xcov_ssq: bad argument"); return CMTFPLS_EINVAL; }
  return run_xcov<double>(X, I, P, Y, ldy, M, S, 0, ws, ws_bytes, (hipStream_t)stream, mean, ssq);
}
size_t cmtfpls_xcov_stats_workspace_bytes(int64_t I, int64_t P, int M) {
  if (I <= 0 || P <= 0 || M <= 0 || M > kXcovMaxResponses) return 0;
  const XcovPlan p = plan_xcov(I, P);
  return (size_t)p.row_blocks * ((size_t)M + 2) * P * sizeof(double);
}
int cmtfpls_xcov_stats_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, double* stats,
                           void* ws, size_t ws_bytes, void* stream) {
  if (!X || !Y || !S || !stats || I <= 0 || P <= 0 || M <= 0 || ldy < M) { set_error("xcov_stats: bad argument"); return CMTFPLS_EINVAL; }
  if (M > kXcovMaxResponses) { set_error("xcov_stats: more than 64 responses; use colstats + xcov"); return CMTFPLS_EUNSUPPORTED; }
  return run_xcov_tile<float>(X, I, P, Y, ldy, M, S, 0, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, stats);
}
int cmtfpls_xcov_stats_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, double* stats,
                           void* ws, size_t ws_bytes, void* stream) {
  if (!X || !Y || !S || !stats || I <= 0 || P <= 0 || M <= 0 || ldy < M) { set_error("xcov_stats: bad argument"); return CMTFPLS_EINVAL; }
  if (M > kXcovMaxResponses) { set_error("xcov_stats: more than 64 responses; use colstats + xcov"); return CMTFPLS_EUNSUPPORTED; }
  return run_xcov_tile<double>(X, I, P, Y, ldy, M, S, 0, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, stats);
}
int cmtfpls_xcov_deflate_f32(float* X, int64_t I, int A, int B, const double* Y, int ldy, int M, const double* t, const double* wA,
                             const double* wB, double* S, double* ssq, void* ws, size_t ws_bytes, void* stream) {
  return run_xcov_deflate<float>(X, I, A, B, Y, ldy, M, t, wA, wB, S, ssq, ws, ws_bytes, (hipStream_t)stream);
}
int cmtfpls_xcov_deflate_f64(double* X, int64_t I, int A, int B, const double* Y, int ldy, int M, const double* t, const double* wA,
                             const double* wB, double* S, double* ssq, void* ws, size_t ws_bytes, void* stream) {
  return run_xcov_deflate<double>(X, I, A, B, Y, ldy, M, t, wA, wB, S, ssq, ws, ws_bytes, (hipStream_t)stream);
}
int cmtfpls_quadform_f64(const double* G, int M, const double* q, const double* q_old, double* out, void* stream) {
  if (!G || !q || !q_old || !out || M <= 0) { set_error("quadform: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(quadform_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, G, M, q, q_old, out);
  return check_launch("quadform");
}

}  // extern "C"
