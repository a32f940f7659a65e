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
#include <type_traits>

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


// ---- round 3: the "j-block" form -------------------------------------------------------------------------------------
//   M[i, r] = sum_j WA[j, r] * ( sum_k X[i, j, k] * WB[k, r] )
// One MFMA = 16 consecutive j-rows of ONE sample x 16 components x 4 k's:
//   A operand  X[i][16 jb + (l & 15)][k]   (the lane's 16-byte vector holds V consecutive k's; element e feeds MFMA e)
//   B operand  WB[k][r = l & 15]           -- a plain loading entry: NO product formed per MFMA (the round-2 form read two
//                                             LDS words and multiplied them for every MFMA), held in REGISTERS for the whole
//                                             kernel when the trailing extent allows (B / 4 doubles per lane), else one LDS read
//   D[(l >> 4) + 4 g][l & 15] = sum_k X[i][16 jb + (l >> 4) + 4 g][k] WB[k][r]; at the end of a j-block the four accumulator
//   entries are multiplied by WA[j][r] and added into ONE running double per lane; the four lane groups are summed by two
//   butterfly steps at the end of the sample.
// A wavefront owns whole samples (no partial rows, fixed summation order) and reads each as 16-row slabs of 16 * B contiguous
// elements (8 KB at 128 x 128 f32) instead of 16 rows 64 KB apart; the next slab's loads are in flight while the current
// one is on the matrix cores.  Shapes: A % 16 == 0, B a multiple of 4 V CH elements (V = 16 / sizeof(T)), R <= 16; anything
// else keeps the tile form above.
template <typename T, int CH, int NPJ, bool BREG>
__global__ __launch_bounds__(256) void mttkrp_jk_kernel(const T* __restrict__ X, int64_t I, int A, int B,
                                                        const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                        double* __restrict__ out, int ldo) {
  constexpr int V = VecOf<T>::N;
  constexpr int RP = 16;
#ifndef CMTFPLS_MTTKRP_NACC
#define CMTFPLS_MTTKRP_NACC 1
#endif
  constexpr int NACC = CMTFPLS_MTTKRP_NACC;
  extern __shared__ double lds[];          // sA[A][16] then sB[B][16], zero padded beyond R
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int jr = lane & 15, kq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int njb = A / 16;
  const int npj = BREG ? NPJ : B / (4 * V * CH);      // chunks of CH patches per j-block
  using XV = Pack<T, V>;
  double breg[BREG ? NPJ * CH : 1][V];
  if (BREG) {
#pragma unroll
    for (int p = 0; p < NPJ * CH; ++p)
#pragma unroll
      for (int e = 0; e < V; ++e) breg[p][e] = sB[(size_t)(p * 4 * V + V * kq + e) * RP + jr];
  }
  const int nchunks = njb * npj, npairs = nchunks / 2;
  const int64_t istep = (int64_t)gridDim.x * 4;
  // Two register buffers in fixed roles (chunks q even / q odd): no buffer is ever copied and no load sits behind a
  // branch, so the wait in front of a chunk's MFMAs covers its own CH loads only (vmcnt(CH)) and the other buffer's CH
  // loads stay in flight behind them.  (A first form that copied `next` into `current`, and a second one that guarded the
  // prefetches with `if (q + 1 < nchunks)`, both made the compiler wait for ALL loads in front of the MFMAs.)  The last
  // prefetch of a sample fetches the first chunk of the wavefront's NEXT sample (its own again when there is none).
  XV b0[CH], b1[CH];
  auto load = [&](XV (&buf)[CH], const T* __restrict__ xs, int q) {
    const int jb = BREG ? q / NPJ : q / npj, c = BREG ? q % NPJ : q - jb * npj;
    const T* __restrict__ xp = xs + (int64_t)jb * 16 * B + c * CH * 4 * V;
#pragma unroll
    for (int p = 0; p < CH; ++p) buf[p] = ld_stream(reinterpret_cast<const XV*>(xp + p * 4 * V));
  };
  int64_t i = (int64_t)blockIdx.x * 4 + wv;
  // the lane's corner of every chunk: row jr of the j-block, columns V kq .. V kq + V - 1 of the patch
  const int64_t lane_off = (int64_t)jr * B + V * kq;
  if (i < I) load(b0, X + i * P + lane_off, 0);
  for (; i < I; i += istep) {
    const T* __restrict__ xs = X + i * P + lane_off;
    const T* __restrict__ xs_next = X + ((i + istep < I) ? i + istep : i) * P + lane_off;
    double s = 0.0;
    // NACC independent accumulator chains (patch p goes to chain p % NACC): a wavefront's next MFMA does not have to wait for
    // its previous one to leave the pipe
    d4m_t accs[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) accs[n] = d4m_t{0.0, 0.0, 0.0, 0.0};
    // CC: the chunk's position inside its j-block when that is known at compile time (register-resident WB), else -1
    auto compute = [&](XV (&buf)[CH], int q, auto cc_tag) {
      constexpr int CC = decltype(cc_tag)::value;
      const int jb = BREG ? q / NPJ : q / npj, c = BREG ? CC : q - jb * npj;
      if (BREG) {
#pragma unroll
        for (int p = 0; p < CH; ++p)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            accs[p % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)buf[p].e[e], breg[(CC < 0 ? 0 : CC) * CH + p][e], accs[p % NACC], 0, 0, 0);
          }
      } else {
        const double* __restrict__ sbc = sB + (size_t)(c * CH * 4 * V + V * kq) * RP + jr;
#pragma unroll
        for (int p = 0; p < CH; ++p)
#pragma unroll
          for (int e = 0; e < V; ++e)
            accs[p % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)buf[p].e[e], sbc[(size_t)(p * 4 * V + e) * RP], accs[p % NACC], 0, 0, 0);
      }
      if (c == npj - 1) {                                   // end of the j-block: fold WA in, start a fresh accumulator
        const double* __restrict__ sa = sA + (size_t)(jb * 16 + kq) * RP + jr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          double d = accs[0][g];
#pragma unroll
          for (int n = 1; n < NACC; ++n) d += accs[n][g];
          s = fma(sa[(size_t)(4 * g) * RP], d, s);
        }
#pragma unroll
        for (int n = 0; n < NACC; ++n) accs[n] = d4m_t{0.0, 0.0, 0.0, 0.0};
      }
    };
    using C0 = std::integral_constant<int, BREG ? 0 : -1>;
    using C1 = std::integral_constant<int, BREG ? (NPJ == 2 ? 1 : 0) : -1>;
    for (int t = 0; t < npairs; ++t) {
      const int q = 2 * t;
      load(b1, xs, q + 1);
      compute(b0, q, C0{});
      const bool more = (q + 2 < nchunks);
      load(b0, more ? xs : xs_next, more ? q + 2 : 0);
      compute(b1, q + 1, C1{});
    }
    if (nchunks & 1) {                                      // an odd last chunk sits in b0; then fetch the next sample's first
      compute(b0, nchunks - 1, C0{});
      load(b0, xs_next, 0);
    }
    s += __shfl_xor(s, 16, kWave);
    s += __shfl_xor(s, 32, kWave);
    if (kq == 0 && jr < R) out[i * ldo + jr] = s;
  }
}


// ---- round 3, second form: "k-row" -- the SAME MFMA count with fully coalesced loads ---------------------------------------
//   M[i, r] = sum_k WB[k, r] * ( sum_j X[i, j, k] * WA[j, r] )
// The matrix cores contract over j: one MFMA = 16 k's of ONE sample x 16 components x 4 consecutive j-rows,
//   A operand  X[i][j0 + (l >> 4)][k]      lane (kk = l & 15, jq = l >> 4) loads the 16 bytes at column 16 V n + V kk of row
//                                          j0 + jq: the 16 lanes of a row read 256 CONTIGUOUS bytes, the four rows of the step
//                                          are contiguous too -- 8 cache lines per load instruction instead of the 64 half-lines
//                                          of the j-block form (whose loads alone cap it at 6.06 TB/s, profiles/r03l_...)
//   B operand  WA[j0 + (l >> 4)][r = l & 15]: ONE LDS word per j-step feeds all NL * V MFMAs of the step
//   D[n][e]    (NL * V = 8 independent accumulators) row (l >> 4) + 4 g <-> k = k0 + 16 V n + V row + e
// After the last j-step the 32 accumulator entries of a lane are multiplied by WB[k][r] (LDS) into one double and the four lane
// groups are summed by two butterfly steps.  Trailing extents beyond NL * 16 V elements are taken in passes of that width
// (256 x 256 f32: two passes over the sample's rows, each reading its half of every row: 512 contiguous bytes per row).
// Shapes: B % (NL * 16 V) == 0, A % (4 * CHJ) == 0 (CHJ = 8 / NL j-steps per register buffer), R <= 16.
template <typename T, int NL>
__global__ __launch_bounds__(256) void mttkrp_kj_kernel(const T* __restrict__ X, int64_t I, int A, int B,
                                                        const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                        double* __restrict__ out, int ldo) {
  constexpr int V = VecOf<T>::N;
  constexpr int RP = 16;
  constexpr int CHJ = 8 / NL;               // j-steps per buffer: 8 loads (8 KB per wavefront) in flight per buffer
  constexpr int KP = NL * 16 * V;           // k's per pass
  extern __shared__ double lds[];          // sA[A][16] then sB[B][16], zero padded beyond R
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kk = lane & 15, jq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int npass = B / KP, cpp = A / (4 * CHJ);          // passes per sample, chunks per pass
  const int nchunks = npass * cpp, npairs = nchunks / 2;
  const int64_t istep = (int64_t)gridDim.x * 4;
  using XV = Pack<T, V>;
  XV b0[CHJ][NL], b1[CHJ][NL];
  // chunk q of a sample: pass q / cpp, j-steps [CHJ (q % cpp), +CHJ)
  auto load = [&](XV (&buf)[CHJ][NL], const T* __restrict__ xs, int q) {
    const int ps = q / cpp, c = q - ps * cpp;
    const T* __restrict__ xp = xs + (int64_t)(c * CHJ * 4) * B + ps * KP;
#pragma unroll
    for (int h = 0; h < CHJ; ++h)
#pragma unroll
      for (int n = 0; n < NL; ++n) buf[h][n] = ld_stream(reinterpret_cast<const XV*>(xp + (int64_t)(h * 4) * B + n * 16 * V));
  };
  int64_t i = (int64_t)blockIdx.x * 4 + wv;
  const int64_t lane_off = (int64_t)jq * B + V * kk;       // row jq of the j-step, columns V kk .. V kk + V - 1 of the load
  if (i < I) load(b0, X + i * P + lane_off, 0);
  for (; i < I; i += istep) {
    const T* __restrict__ xs = X + i * P + lane_off;
    const T* __restrict__ xs_next = X + ((i + istep < I) ? i + istep : i) * P + lane_off;
    double s = 0.0;
    d4m_t acc[NL][V];
#pragma unroll
    for (int n = 0; n < NL; ++n)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[n][e] = d4m_t{0.0, 0.0, 0.0, 0.0};
    auto compute = [&](XV (&buf)[CHJ][NL], int q) {
      const int ps = q / cpp, c = q - ps * cpp;
      const double* __restrict__ sa = sA + (size_t)(c * CHJ * 4 + jq) * RP + kk;   // (kk = l & 15 doubles as the component index r)
#pragma unroll
      for (int h = 0; h < CHJ; ++h) {
        const double wa = sa[(size_t)(h * 4) * RP];
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            acc[n][e] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)buf[h][n].e[e], wa, acc[n][e], 0, 0, 0);
          }
      }
      if (c == cpp - 1) {                                 // end of the pass: fold WB in, start fresh accumulators
        const double* __restrict__ sb = sB + (size_t)(ps * KP) * RP + kk;
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e) {
#pragma unroll
            for (int g = 0; g < 4; ++g) s = fma(sb[(size_t)(n * 16 * V + V * (jq + 4 * g) + e) * RP], acc[n][e][g], s);
            acc[n][e] = d4m_t{0.0, 0.0, 0.0, 0.0};
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


// ---- round 3, third form: the k-row form on v_mfma_f64_4x4x4_4b_f64 for R <= 12 -------------------------------------------------
// With f32 storage the k-row form above is bound by the f64 matrix pipe, not by memory: with half of its MFMAs removed it runs at
// 6.6 TB/s, with all of them at 5.2-5.6 (profiles/r03o_mttkrp_forms.txt) -- and 6 of the 16 columns of every 16x16x4 tile are
// padding when R = 10.  The 4x4x4 form (four independent 4 x 4 x 4 blocks per instruction, same FLOP rate:
// profiles/r03d_mfma_f64_rate.txt) lets the components be taken in groups of FOUR: NG = ceil(R / 4) groups, 12 columns for R = 10.
// Operand layout (probed on the device, profiles/r03e_mfma_f64_4x4_layout.txt): A[blk][i][k] in lane i + 4 blk + 16 k,
// B[blk][k][j] in lane j + 4 blk + 16 k, D[blk][i][j] in lane j + 4 blk + 16 i.  Mapping: contraction index = j-row (l >> 4),
// (i, blk) = the 16 k's of a load (l & 15: the SAME loads and lane map as the k-row form), matrix column = component 4 g + (l & 3):
//   A operand  X[i][j0 + (l >> 4)][k(l & 15)]                     one register, shared by the NG MFMAs of the groups
//   B operand  WA[j0 + (l >> 4)][4 g + (l & 3)]                   NG LDS words per j-step (the same for the four blocks)
//   D[n][e][g] lane l' holds k-row (l' >> 4) + 4 ((l' >> 2) & 3), component 4 g + (l' & 3)
template <typename T, int NL, int NG>
__global__ __launch_bounds__(256) void mttkrp_kj4_kernel(const T* __restrict__ X, int64_t I, int A, int B,
                                                         const double* __restrict__ WA, const double* __restrict__ WB, int R,
                                                         double* __restrict__ out, int ldo) {
  constexpr int V = VecOf<T>::N;
  constexpr int RP = 16;
  constexpr int CHJ = 8 / NL;
  constexpr int KP = NL * 16 * V;
  extern __shared__ double lds[];          // sA[A][16] then sB[B][16], zero padded beyond R
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kk = lane & 15, jq = lane >> 4, jn = lane & 3;
  const int krow = (lane >> 4) + 4 * ((lane >> 2) & 3);   // the k-row this lane's accumulator entries belong to
  const int64_t P = (int64_t)A * B;
  const int npass = B / KP, cpp = A / (4 * CHJ);
  const int nchunks = npass * cpp, npairs = nchunks / 2;
  const int64_t istep = (int64_t)gridDim.x * 4;
  using XV = Pack<T, V>;
  XV b0[CHJ][NL], b1[CHJ][NL];
  auto load = [&](XV (&buf)[CHJ][NL], const T* __restrict__ xs, int q) {
    const int ps = q / cpp, c = q - ps * cpp;
    const T* __restrict__ xp = xs + (int64_t)(c * CHJ * 4) * B + ps * KP;
#pragma unroll
    for (int h = 0; h < CHJ; ++h)
#pragma unroll
      for (int n = 0; n < NL; ++n) buf[h][n] = ld_stream(reinterpret_cast<const XV*>(xp + (int64_t)(h * 4) * B + n * 16 * V));
  };
  int64_t i = (int64_t)blockIdx.x * 4 + wv;
  const int64_t lane_off = (int64_t)jq * B + V * kk;
  if (i < I) load(b0, X + i * P + lane_off, 0);
  for (; i < I; i += istep) {
    const T* __restrict__ xs = X + i * P + lane_off;
    const T* __restrict__ xs_next = X + ((i + istep < I) ? i + istep : i) * P + lane_off;
    double s[NG];
    double acc[NL][V][NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) s[g] = 0.0;
#pragma unroll
    for (int n = 0; n < NL; ++n)
#pragma unroll
      for (int e = 0; e < V; ++e)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[n][e][g] = 0.0;
    auto compute = [&](XV (&buf)[CHJ][NL], int q) {
      const int ps = q / cpp, c = q - ps * cpp;
      const double* __restrict__ sa = sA + (size_t)(c * CHJ * 4 + jq) * RP + jn;
#pragma unroll
      for (int h = 0; h < CHJ; ++h) {
        double wa[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) wa[g] = sa[(size_t)(h * 4) * RP + 4 * g];
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const double a = (double)buf[h][n].e[e];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[n][e][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, wa[g], acc[n][e][g], 0, 0, 0);
          }
      }
      if (c == cpp - 1) {                                 // end of the pass: fold WB in, start fresh accumulators
        const double* __restrict__ sb = sB + (size_t)(ps * KP + V * krow) * RP + jn;
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int e = 0; e < V; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              s[g] = fma(sb[(size_t)(n * 16 * V + e) * RP + 4 * g], acc[n][e][g], s[g]);
              acc[n][e][g] = 0.0;
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
#pragma unroll
    for (int g = 0; g < NG; ++g) {                        // the 16 lanes that share (l & 3): four butterfly steps
      double v = s[g];
      v += __shfl_xor(v, 4, kWave);
      v += __shfl_xor(v, 8, kWave);
      v += __shfl_xor(v, 16, kWave);
      v += __shfl_xor(v, 32, kWave);
      if (lane < 4 && 4 * g + jn < R) out[i * ldo + 4 * g + jn] = v;
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
#ifndef CMTFPLS_MTTKRP_TILE_ONLY
  {
    // the j-block form (round 3): whole samples per wavefront, WB as the plain B operand (registers or LDS)
    constexpr int V = 16 / (int)sizeof(T);
    const size_t lds1 = (size_t)(A + B) * 16 * sizeof(double);
    const bool aligned = (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (R <= 16 && A % 16 == 0 && aligned && lds1 <= kMttkrpLdsMax) {
#ifndef CMTFPLS_MTTKRP_JK_GRID
#define CMTFPLS_MTTKRP_JK_GRID 2048
#endif
      int grid = (int)((I + 3) / 4);
      if (grid > CMTFPLS_MTTKRP_JK_GRID) grid = CMTFPLS_MTTKRP_JK_GRID;
      const dim3 g(grid), b(256);
#define JKL(CHH, NPJJ, BRG)                                                                                                   \
  do {                                                                                                                        \
    if (lds1 > 64 * 1024)                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mttkrp_jk_kernel<T, CHH, NPJJ, BRG>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);                                       \
    hipLaunchKernelGGL((mttkrp_jk_kernel<T, CHH, NPJJ, BRG>), g, b, lds1, st, X, I, A, B, WA, WB, R, out, ldo);                \
    return check_launch("mttkrp");                                                                                            \
  } while (0)
#ifndef CMTFPLS_MTTKRP_NO_KJ
      // the k-row form first (coalesced loads): f32 128 / 256 / ... columns per row (NL = 2), 64 (NL = 1); f64 128+ (NL = 4), 64 (NL = 2)
#define KJL(NLL)                                                                                                              \
  do {                                                                                                                        \
    if (lds1 > 64 * 1024)                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mttkrp_kj_kernel<T, NLL>),                                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);                                       \
    hipLaunchKernelGGL((mttkrp_kj_kernel<T, NLL>), g, b, lds1, st, X, I, A, B, WA, WB, R, out, ldo);                           \
    return check_launch("mttkrp");                                                                                            \
  } while (0)
#define KJ4(NLL, NGG)                                                                                                         \
  do {                                                                                                                        \
    if (lds1 > 64 * 1024)                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mttkrp_kj4_kernel<T, NLL, NGG>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);                                       \
    hipLaunchKernelGGL((mttkrp_kj4_kernel<T, NLL, NGG>), g, b, lds1, st, X, I, A, B, WA, WB, R, out, ldo);                     \
    return check_launch("mttkrp");                                                                                            \
  } while (0)
#define KJ4N(NLL)                                                                                                             \
  do {                                                                                                                        \
    if (R <= 4) KJ4(NLL, 1); else if (R <= 8) KJ4(NLL, 2); else if (R <= 12) KJ4(NLL, 3); else KJ4(NLL, 4);                   \
  } while (0)
      {
        constexpr int NLW = (V == 4) ? 2 : 4, NLH = NLW / 2;            // loads per j-step: 128 columns per pass, or 64
        const bool wide = B % (NLW * 16 * V) == 0 && A % (4 * (8 / NLW)) == 0;
        const bool half = B % (NLH * 16 * V) == 0 && A % (4 * (8 / NLH)) == 0;
#ifndef CMTFPLS_MTTKRP_NO_KJ4
        // f32 storage, R <= 12: the 4x4x4 form (the 16x16x4 one is matrix-pipe bound there; f64 storage is memory bound either way)
#ifndef CMTFPLS_MTTKRP_KJ4_MAXR
#define CMTFPLS_MTTKRP_KJ4_MAXR 12
#endif
        if (V == 4 && R <= CMTFPLS_MTTKRP_KJ4_MAXR) {
          if (wide) KJ4N(NLW);
          if (half) KJ4N(NLH);
        }
#endif
        if (wide) KJL(NLW);
        if (half) KJL(NLH);
      }
#undef KJ4N
#undef KJ4
#undef KJL
#endif
      const int npatch = (B % (4 * V) == 0) ? B / (4 * V) : 0;          // patches of 4 V columns per j-row
#ifndef CMTFPLS_MTTKRP_NOBREG
      if (npatch == 8) JKL(8, 1, true);                                 // 128 f32 / 64 f64: WB in 32 / 16 doubles per lane
      if (npatch == 4) JKL(4, 1, true);
      if (npatch == 16 && V == 2) JKL(8, 2, true);                      // 128 f64: 32 doubles per lane
#endif
      if (npatch > 0 && npatch % 8 == 0) JKL(8, 1, false);              // 256 f32 and beyond: WB from LDS, one read per MFMA
      if (npatch > 0 && npatch % 4 == 0) JKL(4, 1, false);
#undef JKL
    }
  }
#endif
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
