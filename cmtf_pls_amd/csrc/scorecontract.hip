// Score and the contraction with that score in ONE read of X (round 3):
//   t[i] = sum_c X[i, c] w[c] - shift - sub_own[i],    c[i] = alpha (t[i] + add_other[i]),    Z[col] = sum_i c[i] X[i, col]
// (w[c] = wA[c / B] wB[c % B]; shift, sub_own, add_other nullable = 0), i.e. t = X w (multi_mode_dot, tpls.py:97-99) and
// Z = X x_0 c (np.einsum, tpls.py:83) with a score the same pass just formed.  sub_own = T[:, :a] (w_j^T w_a)_j turns X_0 w into the
// score of the implicitly deflated X_a; add_other = the other coupled blocks' scores and alpha = 1 / blocks make c the
// block-AVERAGED score (cmtf.py:120) the deflation uses -- so Z = X_0^T t_a for the block that is read last.
// Why it exists: the cross-covariance loop that never writes X (engine.FitRun._finish_xcov_nowrite) needs, per component, the
// final score t_a and the down-date vector X_{a+1}^T yhat with yhat = T b.  yhat is a combination of the scores, so
// X_0^T yhat = sum_j b_j (X_0^T t_j): with r_j = X_0^T t_j kept from the pass that produced t_j, the second read of X per component
// disappears.  t_a = s_a - sum_{j<a} t_j (w_j^T w_a) with s_a = X_0 w_a, hence r_a = X_0^T s_a - sum_{j<a} r_j (w_j^T w_a): the one
// thing that needs X is p = X_0^T s_a -- this kernel.  (One block only: with coupled blocks the deflation uses the block-averaged
// score, whose contraction with block b is not something a pass over block b alone can form.)
//
// One 1024-thread workgroup per CU walks rows r = blockIdx.x, + gridDim.x, ...; a lane owns the same NV vectors of 16 bytes
// in every row -- their loading products and NV * V f64 accumulators of Z in registers; the row's dot product is a block sum
// (one barrier per row), the next row's loads are in flight meanwhile (always issued: a row index clamped to the last row
// instead of a branch, which would make the compiler wait for every load).  One partial row of Z per workgroup, added in index
// order by reduce_rows_kernel.  shift (device, nullable): X is uncentred, t = X w - mean^T w.  csum (nullable): sum_i c[i], what the
// rank-one correction X_c^T c = X^T c - (1^T c) mean of an uncentred X needs (a 65536-element sum_kernel launch otherwise).
// Shapes: B % V == 0, no missing values; rows of at most 1024 * V * 4 (f32) / 1024 * V * 8 (f64) elements = 16384 in this form, up to
// 16 x 16384 (f32) / 16 x 8192 (f64) elements in the split form further down (round 4: the north-star 256 x 256 row).
#include "common.hpp"

namespace cmtfpls {

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st);

constexpr int kScGrid = 256;

// Sum over each row of 16 lanes with data-parallel-primitive moves (no LDS, 2 registers): lane ^ 1, lane ^ 2, then the two
// mirrors (within 8, within 16) -- after the quad steps all lanes of a quad agree, so a mirror adds the other quads.  Every
// lane of the row ends with the same bits.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const uint64_t b = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move<0xB1>(v);      // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);      // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x141>(v);     // row_half_mirror
  v += dpp_move<0x140>(v);     // row_mirror
  return v;
}

// KC: 1024 * V is a multiple of B, so a lane meets the same mode-2 index in all its vectors -- one set of wB values, not NV.
// WL (the 8-vector rows of f64 storage, where 2 x 32 registers of rows in flight + 32 of accumulators leave no room): KC, every
// vector exists (P = NV * 1024 * V) and A <= kScLdsA -- the mode-1 loadings are read from LDS per row instead of held, and the
// column offsets are compile-time multiples of the stride.
constexpr int kScLdsA = 2048;
template <typename T, int NV, bool KC, bool WL = false>
__global__ __launch_bounds__(1024) void score_contract_rows_kernel(const T* __restrict__ X, int64_t I, unsigned P, int B,
                                                                  const double* __restrict__ wA, const double* __restrict__ wB,
                                                                  const double* __restrict__ shift, const double* __restrict__ sub_own,
                                                                  const double* __restrict__ add_other, double alpha,
                                                                  double* __restrict__ t, double* __restrict__ part,
                                                                  double* __restrict__ csum_part) {
  __shared__ double red[2][16];
  __shared__ double wls[WL ? kScLdsA : 1];
  constexpr int V = VecOf<T>::N;
  using VT = Pack<T, V>;
  constexpr unsigned stride = 1024u * V;
  const unsigned c0 = threadIdx.x * V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double wa[WL ? 1 : NV], wb[KC ? 1 : NV][V], acc[NV][V];
  bool ok[NV];
  const unsigned j0 = c0 / (unsigned)B, jstep = stride / (unsigned)B;     // WL: vector n of this lane is in mode-1 slice j0 + n * jstep
  if constexpr (WL) {
    for (unsigned j = threadIdx.x; j < NV * jstep; j += 1024) wls[(j % jstep) * NV + j / jstep] = wA[j];   // [slice within the stride][vector]
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const unsigned c = c0 + n * stride;
    ok[n] = WL || c < P;
    const unsigned cg = ok[n] ? c : 0;
    if constexpr (!WL) wa[n] = ok[n] ? wA[cg / (unsigned)B] : 0.0;      // B % V == 0: one j for the whole vector
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (!KC || n == 0) wb[KC ? 0 : n][e] = wB[cg % (unsigned)B + e];
      acc[n][e] = 0.0;
    }
  }
  const double sh = shift ? shift[0] : 0.0;
  unsigned col[WL ? 1 : NV];                                  // a vector that does not exist reads column 0 (weight 0): in bounds
  if constexpr (!WL) {
#pragma unroll
    for (int n = 0; n < NV; ++n) col[n] = ok[n] ? c0 + n * stride : 0u;
  }
  const int64_t step = gridDim.x;
  int64_t r = blockIdx.x;
  VT bufA[NV], bufB[NV];                                     // fixed roles, never copied (a copy would wait for the loads)
  int parity = 0;
  double csum = 0.0;
  auto load = [&](VT (&buf)[NV], int64_t row) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if constexpr (WL) {                                    // uniform base per vector + ONE lane offset: no per-vector address registers
        const uint64_t o = (uint64_t)(row * (int64_t)P + (int64_t)n * stride);      // made visibly uniform: scalar base, one lane offset
        const uint64_t ou = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(o >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)o);
        buf[n] = ld_stream(reinterpret_cast<const VT*>(X + ou + c0));
      } else {
        buf[n] = ld_stream(reinterpret_cast<const VT*>(X + row * (int64_t)P + col[n]));
      }
    }
  };
  auto use = [&](VT (&buf)[NV], int64_t row) {
    double d = 0.0;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      double dn = 0.0;
#pragma unroll
      for (int e = 0; e < V; ++e) dn = fma((double)buf[n].e[e], wb[KC ? 0 : n][e], dn);
      d = fma(WL ? wls[j0 * NV + n] : wa[WL ? 0 : n], dn, d);       // (WL: the lane's NV loadings are contiguous in LDS)
    }
    d = wave_sum(d);
    if (lane == 0) red[parity][wv] = d;
    __syncthreads();
    double ti = row16_sum(red[parity][lane & 15]) - sh;          // the 16 wavefronts' partial sums, one per lane of a row
    parity ^= 1;
    if (sub_own) ti -= sub_own[row];
    if (threadIdx.x == 0) t[row] = ti;
    if (add_other) ti += add_other[row];
    ti *= alpha;
    csum += ti;                                                // (every lane carries the same running sum of the weights)
    if constexpr (sizeof(T) == 4) {                          // convert again rather than keep 16 f64 copies alive across the barrier
#pragma unroll
      for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < V; ++e) asm volatile("" : "+v"(buf[n].e[e]));
    }
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[n][e] = fma(ti, (double)buf[n].e[e], acc[n][e]);
  };
  const int64_t last = I - 1;
  if (r < I) load(bufA, r);
  while (r < I) {
    const int64_t r1 = r + step;
    load(bufB, r1 < I ? r1 : last);                          // always issued (a clamped row, not a branch around the loads)
    use(bufA, r);
    if (r1 >= I) break;
    const int64_t r2 = r1 + step;
    load(bufA, r2 < I ? r2 : last);
    use(bufB, r1);
    r = r2;
  }
  if (csum_part && threadIdx.x == 0) csum_part[blockIdx.x] = csum;
  double* __restrict__ prow = part + (int64_t)blockIdx.x * P + c0;
#pragma unroll
  for (int n = 0; n < NV; ++n)
    if (ok[n]) {
#pragma unroll
      for (int e = 0; e < V; ++e) prow[n * stride + e] = acc[n][e];
    }
}

// ---- rows LONGER than one workgroup's registers (round 4): the row split over G workgroups ------------------------------------
// P f64 accumulators per row stream are what limits the form above (65536 of them = the whole register file of a CU).  Here a row
// is cut into G column slabs of NV * 1024 * V elements and G workgroups with CONSECUTIVE block indices (dispatched together, one
// workgroup per CU, grid <= CUs: co-resident) walk the same rows, each holding its own slab and its slab's accumulators.  The one
// thing they owe each other is the row's dot product: workgroup g publishes its partial sum with ONE 8-byte agent-scope store into
// xch[row * G + g] -- the value is its own flag (slots are preset to an all-ones pattern no sum produces), so there is no release
// fence and no L2 write-back -- and all G workgroups add the G slots in index order LAG rows later (identical bits in every
// partner), by which time the stores have long landed: the exchange costs no stall, only LAG + 2 row buffers of registers.
// One barrier per row as before; wavefront 0 polls (a bounded spin: if a partner never shows up the score becomes NaN and the grid
// still drains).  Row stream s = blockIdx.x / G walks rows s, s + S, ...; slab g = blockIdx.x % G; t and csum come from slab 0.
constexpr unsigned long long kXchEmpty = ~0ull;
constexpr int kSplitMaxG = 16;
constexpr int kSplitSpinLimit = 1 << 21;

__global__ __launch_bounds__(256) void xch_preset_kernel(unsigned long long* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = kXchEmpty;
}

template <typename T, int NV, bool KC, int LAG>
__global__ __launch_bounds__(1024) void score_contract_split_kernel(const T* __restrict__ X, int64_t I, unsigned P, int B, int G,
                                                                   const double* __restrict__ wA, const double* __restrict__ wB,
                                                                   const double* __restrict__ shift, const double* __restrict__ sub_own,
                                                                   const double* __restrict__ add_other, double alpha,
                                                                   double* __restrict__ t, double* __restrict__ part,
                                                                   double* __restrict__ csum_part, unsigned long long* xch) {
  __shared__ double red[2][16];
  __shared__ double tis[2];
  constexpr int V = VecOf<T>::N;
  using VT = Pack<T, V>;
  constexpr unsigned stride = 1024u * V;
  constexpr int NB = LAG + 2;                                 // row k + 1 arriving, row k just summed, ..., row k - LAG being accumulated
  const int g = (int)(blockIdx.x % (unsigned)G);
  const int64_t s = blockIdx.x / (unsigned)G, S = gridDim.x / (unsigned)G;
  const unsigned c0 = (unsigned)g * (NV * stride) + threadIdx.x * V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double wa[NV], wb[KC ? 1 : NV][V], acc[NV][V];
  bool ok[NV];
  unsigned col[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const unsigned c = c0 + n * stride;
    ok[n] = c < P;
    const unsigned cg = ok[n] ? c : 0;
    col[n] = cg;                                               // a vector that does not exist reads column 0 with weight 0: in bounds
    wa[n] = ok[n] ? wA[cg / (unsigned)B] : 0.0;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (!KC || n == 0) wb[KC ? 0 : n][e] = wB[cg % (unsigned)B + e];
      acc[n][e] = 0.0;
    }
  }
  const double sh = shift ? shift[0] : 0.0;
  const int64_t nrows = s < I ? (I - s + S - 1) / S : 0;      // the same for the G partners of a stream
  VT buf[NB][NV];
  double csum = 0.0;
  bool dead = false;
  int par = 0;
  auto load = [&](VT (&b)[NV], int64_t row) {
#pragma unroll
    for (int n = 0; n < NV; ++n) b[n] = ld_stream(reinterpret_cast<const VT*>(X + row * (int64_t)P + col[n]));
  };
  if (nrows > 0) load(buf[0], s);
  const int64_t steps = nrows > 0 ? nrows + LAG : 0;           // (a stream without rows makes no step: nothing to clamp a load to)
  for (int64_t k0 = 0; k0 < steps; k0 += NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int64_t k = k0 + j;
      if (k < steps) {                                         // (uniform)
        const int64_t kn = k + 1 < nrows ? k + 1 : nrows - 1;  // always issued, a clamped row: the buffer it lands in is free either way
        load(buf[(j + 1) % NB], s + kn * S);
        if (k < nrows) {                                       // this slab's share of row k's dot product
          double d = 0.0;
#pragma unroll
          for (int n = 0; n < NV; ++n) {
            double dn = 0.0;
#pragma unroll
            for (int e = 0; e < V; ++e) dn = fma((double)buf[j][n].e[e], wb[KC ? 0 : n][e], dn);
            d = fma(wa[n], dn, d);
          }
          d = wave_sum(d);
          if (lane == 0) red[par][wv] = d;
        }
        const int64_t kc = k - LAG;                            // the row whose G partial sums were published LAG steps ago
        if (kc >= 0 && wv == 0) {
          const unsigned long long* slot = xch + (s + kc * S) * (int64_t)G + (lane < G ? lane : 0);
          unsigned long long bits = kXchEmpty;
          if (!dead) {
            int spins = 0;
            for (;;) {
              bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__all(bits != kXchEmpty)) break;
              if (++spins > kSplitSpinLimit) { dead = true; break; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
          double tot = 0.0;
          for (int gg = 0; gg < G; ++gg) tot += __shfl(__longlong_as_double((long long)bits), gg, kWave);   // index order: same bits in every partner
          if (dead) tot = __longlong_as_double(0x7FF8000000000000ll);
          if (lane == 0) tis[par] = tot;
        }
        __syncthreads();
        if (k < nrows) {
          const double tot = row16_sum(red[par][lane & 15]);
          if (threadIdx.x == 0) {
            unsigned long long bits = (unsigned long long)__double_as_longlong(tot);
            if (bits == kXchEmpty) bits = 0x7FF8000000000000ull;
            __hip_atomic_store(xch + (s + k * S) * (int64_t)G + g, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (kc >= 0) {
          const int64_t row = s + kc * S;
          double ti = tis[par] - sh;
          if (sub_own) ti -= sub_own[row];
          if (g == 0 && threadIdx.x == 0) t[row] = ti;
          if (add_other) ti += add_other[row];
          ti *= alpha;
          csum += ti;
          VT (&bc)[NV] = buf[(j + NB - LAG) % NB];
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
              for (int e = 0; e < V; ++e) asm volatile("" : "+v"(bc[n].e[e]));
          }
#pragma unroll
          for (int n = 0; n < NV; ++n)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[n][e] = fma(ti, (double)bc[n].e[e], acc[n][e]);
        }
        par ^= 1;
      }
    }
  }
  if (csum_part && g == 0 && threadIdx.x == 0) csum_part[s] = csum;
  double* __restrict__ prow = part + s * (int64_t)P;
#pragma unroll
  for (int n = 0; n < NV; ++n)
    if (ok[n]) {
#pragma unroll
      for (int e = 0; e < V; ++e) prow[col[n] + e] = acc[n][e];
    }
}

static int split_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}

template <typename T>
static int run_score_contract(const T* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* shift,
                              const double* sub_own, const double* add_other, double alpha, double* t, double* Z, double* csum,
                              void* ws, size_t ws_bytes, hipStream_t st) {
  if (!X || !wA || !wB || !t || !Z || I <= 0 || A <= 0 || B <= 0) { set_error("score_contract: bad argument"); return CMTFPLS_EINVAL; }
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t P = (int64_t)A * B;
  const int64_t stride = (int64_t)1024 * V;
  const int nv = (int)((P + stride - 1) / stride);
  constexpr int kMaxNV = (V == 4) ? 4 : 8;
  if ((B % V) != 0 || P < stride / 2 || P >= ((int64_t)1 << 31) || (reinterpret_cast<uintptr_t>(X) & 15) != 0) {
    set_error("score_contract: shape outside the row-in-registers form; use score + mode0_contract");
    return CMTFPLS_EUNSUPPORTED;
  }
  const bool kc = (stride % B) == 0;
  if (nv > kMaxNV) {                                          // the row split over G workgroups
    // vectors per lane and slab: 4 (16384 f32 / 8192 f64 elements per workgroup); 2 for f32 rows whose lanes meet different mode-2
    // indices in their vectors (4 sets of wB would spill: 0.62 ms against 0.54 ms of two passes at 8192 x 250 x 200)
    const int NVS = (sizeof(T) == 4 && !kc) ? 2 : 4;
    const int G = (nv + NVS - 1) / NVS;
    const int cus = split_cus() < kScGrid ? split_cus() : kScGrid;
    const int S = (int)(I < cus / G ? I : cus / G);           // row streams: every one has a row (the kernel's clamped loads rely on it)
    if (G > kSplitMaxG || S < 1) {
      set_error("score_contract: row beyond 16 workgroups' registers; use score + mode0_contract");
      return CMTFPLS_EUNSUPPORTED;
    }
    const size_t need = ((size_t)S * (P + 1) + (size_t)I * G) * sizeof(double);
    if (!ws || ws_bytes < need) { set_error("score_contract: workspace too small"); return CMTFPLS_EWORKSPACE; }
    double* part = static_cast<double*>(ws);
    double* csum_part = part + (size_t)S * P;
    unsigned long long* xch = reinterpret_cast<unsigned long long*>(csum_part + S);
    hipLaunchKernelGGL(xch_preset_kernel, dim3((unsigned)(((int64_t)I * G + 2047) / 2048)), dim3(256), 0, st, xch, (int64_t)I * G);
#define SPL(NVV, KCC, LG)                                                                                                          \
  hipLaunchKernelGGL((score_contract_split_kernel<T, NVV, KCC, LG>), dim3(S * G), dim3(1024), 0, st, X, I, (unsigned)P, B, G, wA, wB, \
                     shift, sub_own, add_other, alpha, t, part, csum ? csum_part : nullptr, xch)
    // rows between publishing a partial sum and using the total: as many as the registers hold without spilling (measured at
    // 256 x 256, profiles/r04m_split_rows.txt: f32 1 row 1.37 ms / 2 rows (spills) 1.73 ms at 32768 rows; f64 1.39 / 1.33 ms at 16384)
    if constexpr (sizeof(T) == 8) {
      if (kc) SPL(4, true, 2);
      else SPL(4, false, 1);
    } else {
      if (kc) SPL(4, true, 1);
      else SPL(2, false, 2);
    }
#undef SPL
    launch_reduce_rows(part, S, P, Z, st);
    if (csum) launch_reduce_rows(csum_part, S, 1, csum, st);
    return check_launch("score_contract (split rows)");
  }
  const int grid = (int)(I < kScGrid ? I : kScGrid);
  if (!ws || ws_bytes < (size_t)grid * (P + 1) * sizeof(double)) { set_error("score_contract: workspace too small"); return CMTFPLS_EWORKSPACE; }
  double* part = static_cast<double*>(ws);
  double* csum_part = csum ? part + (size_t)grid * P : nullptr;
#define SCL(NVV)                                                                                                                  \
  do {                                                                                                                            \
    if (kc) hipLaunchKernelGGL((score_contract_rows_kernel<T, NVV, true>), dim3(grid), dim3(1024), 0, st, X, I, (unsigned)P, B, wA, wB, shift, sub_own, add_other, alpha, t, part, csum_part); \
    else hipLaunchKernelGGL((score_contract_rows_kernel<T, NVV, false>), dim3(grid), dim3(1024), 0, st, X, I, (unsigned)P, B, wA, wB, shift, sub_own, add_other, alpha, t, part, csum_part); \
  } while (0)
  if (nv <= 1) SCL(1);
  else if (nv <= 2) SCL(2);
  else if (nv <= 4) SCL(4);
  else if constexpr (kMaxNV >= 8) {
    if (kc && P == 8 * stride && A <= kScLdsA)
      hipLaunchKernelGGL((score_contract_rows_kernel<T, 8, true, true>), dim3(grid), dim3(1024), 0, st, X, I, (unsigned)P, B, wA, wB, shift, sub_own, add_other, alpha, t, part, csum_part);
    else SCL(8);
  }
#undef SCL
  launch_reduce_rows(part, grid, P, Z, st);
  if (csum) launch_reduce_rows(csum_part, grid, 1, csum, st);
  return check_launch("score_contract");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
size_t cmtfpls_score_contract_workspace_bytes(int64_t I, int64_t P) {
  if (I <= 0 || P <= 0) return 0;
  return ((size_t)(I < kScGrid ? I : kScGrid) * (size_t)(P + 1) + (size_t)I * kSplitMaxG) * sizeof(double);   // partial rows of Z | csum | exchange slots
}
int cmtfpls_score_contract_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* shift,
                               const double* sub_own, const double* add_other, double alpha, double* t, double* Z, double* csum,
                               void* ws, size_t ws_bytes, void* stream) {
  return run_score_contract<float>(X, I, A, B, wA, wB, shift, sub_own, add_other, alpha, t, Z, csum, ws, ws_bytes, (hipStream_t)stream);
}
int cmtfpls_score_contract_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* shift,
                               const double* sub_own, const double* add_other, double alpha, double* t, double* Z, double* csum,
                               void* ws, size_t ws_bytes, void* stream) {
  return run_score_contract<double>(X, I, A, B, wA, wB, shift, sub_own, add_other, alpha, t, Z, csum, ws, ws_bytes, (hipStream_t)stream);
}
}
