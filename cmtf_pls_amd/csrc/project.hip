// transform / predict for inputs WITH missing values (tpls.py:128-142, 151-165 and cmtf.py:143-177, 180-210 with
// miss_mmodedot, missingvals.py:23-38): the reference centres a copy of X and then, R times, projects every sample on the
// component's loadings (per-row rescale P / n_observed), averages the blocks' scores when the blocks are coupled
// (np.average, cmtf.py:155,206) and deflates the copy -- R read + write passes (cmtfpls_score_deflate_*).  The samples are
// independent, so here a workgroup takes one SAMPLE -- its row of the block, or of BOTH coupled blocks -- keeps it in
// registers, and runs the whole sequence on it: centring (x - mean, rounded to the storage type as the stored copy would
// be), the observation counts, and for a = 0..R-1 the masked scores, the rescale, the block average and the rank-one
// deflations (rounded to the storage type again, as the write-back would).  X is read ONCE and never written; the
// arithmetic per element and the summation order (per lane over its vectors, butterfly wave sum, wavefronts in index
// order) are those of the sequential kernels.  The loadings of all R components sit in LDS, component-major.
//
// Shapes (round 3): a lane holds up to 16 vectors of 16 bytes of its sample.  256-thread workgroups take rows of up to
// 256 * 16 vectors (128 x 128 f32, 64 x 128 f64), four samples in flight per CU; 1024-thread workgroups (one sample per
// CU, 128 registers per lane) take rows of up to 1024 * 16 vectors (256 x 256 f32 = BASELINE configs[4], 256 x 128 f64).
// Two coupled blocks share a 256-thread workgroup when the second (shorter) one needs at most 4 vectors per lane (the
// I x 512 matrix block of BASELINE configs[2]: one).  In every case the trailing extent B of a block must divide the
// workgroup stride (NT * V elements: "k constant per lane", every power-of-two B); other shapes, three or more coupled
// blocks, and blocks of different storage types keep the passes.
#include "common.hpp"

namespace cmtfpls {

template <typename T>
struct ProjBlock {
  const T* X;            // I x (A * B), uncentred
  const double* WA;      // A x R
  const double* WB;      // B x R
  const double* mean;    // A * B (nullable)
  int A, B;
};

// One block's share of the sample: NV vectors per lane.  The workgroup stride (NT * V elements) is a multiple of B, so
// every vector a lane owns has the same k = c % B: the V entries of wB it needs are read from LDS once per component,
// and j advances by the uniform dj = stride / B.
template <typename T, int NT, int NV>
struct RowPart {
  static constexpr int V = VecOf<T>::N;
  static constexpr unsigned stride = NT * V;
  using VT = Pack<T, V>;
  VT x[NV > 0 ? NV : 1];
  unsigned c0, P;
  int j0, k0, dj;

  __device__ __forceinline__ void init(int A, int B) {
    P = (unsigned)A * (unsigned)B;
    c0 = threadIdx.x * V;
    j0 = (int)(c0 / (unsigned)B);
    k0 = (int)(c0 % (unsigned)B);
    dj = (int)(stride / (unsigned)B);
  }
  __device__ __forceinline__ void load(const T* X, int64_t row) {
    // scalar row base (opaque to loop strength reduction, which otherwise keeps one 64-bit running address per vector in
    // VGPRs): every access is SGPR base + one shared VGPR offset
    const uint64_t rb = reinterpret_cast<uint64_t>(X + row * (int64_t)P);
    const uint32_t rb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rb >> 32));
    const uint32_t rb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb);
    const T* __restrict__ xr = reinterpret_cast<const T*>(((uint64_t)rb_hi << 32) | (uint64_t)rb_lo);
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (c0 + n * stride < P) x[n] = ld_stream(reinterpret_cast<const VT*>((xr + (int64_t)n * stride) + c0));
  }
  // x <- (T)(x - mean); returns this lane's count of observed entries
  __device__ __forceinline__ double centre_count(const double* __restrict__ mean) {
    double cnt = 0.0;
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (c0 + n * stride < P) {
        unsigned cm = c0 + n * stride;
        asm volatile("" : "+v"(cm));                                  // one offset at a time, not NV of them kept per lane
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const double mu = mean ? mean[cm + e] : 0.0;                // L2-resident; a register copy would cost 2 * NV * V VGPRs
          const T nv = (T)((double)x[n].e[e] - mu);                   // the centred copy, in the storage type (NaN stays NaN)
          x[n].e[e] = nv;
          cnt += (nv == nv) ? 1.0 : 0.0;
        }
        if (NV > 4) __builtin_amdgcn_sched_barrier(0);
      }
    return cnt;
  }
  // keep the row opaque between the phases: otherwise its f64 conversions are kept alive for reuse (2 VGPRs per element)
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
      for (int e = 0; e < V; ++e) asm volatile("" : "+v"(x[n].e[e]));
  }
  // this lane's share of sum over observed c of x[c] * wA[c / B] * wB[c % B]
  __device__ __forceinline__ double dot(const double* __restrict__ sAa, const double (&wb)[V]) {
    double acc = 0.0;
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (c0 + n * stride < P) {
        double d = 0.0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T xv = x[n].e[e];
          d = fma((xv == xv) ? (double)xv : 0.0, wb[e], d);
        }
        int jn = j0 + n * dj;
        asm volatile("" : "+v"(jn));                               // no per-vector LDS address kept (and advanced) across the loops
        acc = fma(sAa[jn], d, acc);
        if (NV > 4) __builtin_amdgcn_sched_barrier(0);            // keep live temporaries low: the row owns the VGPRs
      }
    return acc;
  }
  // x <- (T)(x - ti * wA[j] * wB[k])      (tpls.py:142 / cmtf.py:173-176 on the stored type)
  __device__ __forceinline__ void deflate(double ti, const double* __restrict__ sAa, const double (&wb)[V]) {
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (c0 + n * stride < P) {
        int jn = j0 + n * dj;
        asm volatile("" : "+v"(jn));
        const double tw = ti * sAa[jn];
#pragma unroll
        for (int e = 0; e < V; ++e) x[n].e[e] = (T)fma(-tw, wb[e], (double)x[n].e[e]);
        if (NV > 4) __builtin_amdgcn_sched_barrier(0);
      }
  }
};

// NV1 == 0: one block (tPLS, or ctPLS with a single block).  NV1 > 0: two coupled blocks, score = mean of the two.
// (the one-block 16-vector instance at 256 threads is held to 128 registers -- four workgroups per CU instead of three;
// left alone the allocator takes 132)
template <typename T, int NT, int NV0, int NV1>
__global__ __launch_bounds__(NT, (NT == 256 && NV0 == 16 && NV1 == 0) ? 4 : 1) void project_rows_kernel(ProjBlock<T> b0, ProjBlock<T> b1, int64_t I, int R,
                                                         double* __restrict__ scores, int ld, const int64_t* __restrict__ rows) {
  extern __shared__ double lds[];                    // sA0[R][A0] | sB0[R][B0] | sA1[R][A1] | sB1[R][B1]
  constexpr int NW = NT / 64;
  constexpr int V = VecOf<T>::N;
  constexpr bool TWO = NV1 > 0;
  __shared__ double red[2][NW][2];
  double* sA0 = lds;
  double* sB0 = sA0 + (size_t)R * b0.A;
  double* sA1 = sB0 + (size_t)R * b0.B;
  double* sB1 = sA1 + (TWO ? (size_t)R * b1.A : 0);
  for (int idx = threadIdx.x; idx < R * b0.A; idx += NT) { const int a = idx / b0.A, j = idx % b0.A; sA0[idx] = b0.WA[(int64_t)j * R + a]; }
  for (int idx = threadIdx.x; idx < R * b0.B; idx += NT) { const int a = idx / b0.B, k = idx % b0.B; sB0[idx] = b0.WB[(int64_t)k * R + a]; }
  if (TWO) {
    for (int idx = threadIdx.x; idx < R * b1.A; idx += NT) { const int a = idx / b1.A, j = idx % b1.A; sA1[idx] = b1.WA[(int64_t)j * R + a]; }
    for (int idx = threadIdx.x; idx < R * b1.B; idx += NT) { const int a = idx / b1.B, k = idx % b1.B; sB1[idx] = b1.WB[(int64_t)k * R + a]; }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  RowPart<T, NT, NV0> p0;
  RowPart<T, NT, NV1> p1;
  p0.init(b0.A, b0.B);
  if (TWO) p1.init(b1.A, b1.B);
  __syncthreads();
  int parity = 0;
  // rows != null: only the listed samples (the ones WITH a missing value inside a batch whose complete samples keep their
  // one-pass MTTKRP scores); I is then the length of the list and scores is still indexed by the sample
  for (int64_t it = blockIdx.x; it < I; it += gridDim.x) {
    const int64_t row = rows ? rows[it] : it;
    p0.load(b0.X, row);
    if (TWO) p1.load(b1.X, row);
    double cnt0 = wave_sum(p0.centre_count(b0.mean));
    double cnt1 = TWO ? wave_sum(p1.centre_count(b1.mean)) : 0.0;
    if (lane == 0) { red[parity][wv][0] = cnt0; if (TWO) red[parity][wv][1] = cnt1; }
    __syncthreads();
    double rowcnt0 = 0.0, rowcnt1 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { rowcnt0 += red[parity][w][0]; if (TWO) rowcnt1 += red[parity][w][1]; }
    parity ^= 1;
    bool bad = false;                                              // a NaN score makes every later score of the sample NaN
#pragma unroll 1
    for (int a = 0; a < R; ++a) {
      const double* __restrict__ sAa0 = sA0 + (size_t)a * b0.A;
      const double* __restrict__ sAa1 = sA1 + (TWO ? (size_t)a * b1.A : 0);
      double wb0[V], wb1[V];
#pragma unroll
      for (int e = 0; e < V; ++e) wb0[e] = sB0[(size_t)a * b0.B + p0.k0 + e];
      if (TWO) {
#pragma unroll
        for (int e = 0; e < V; ++e) wb1[e] = sB1[(size_t)a * b1.B + p1.k0 + e];
      }
      p0.opaque();
      double acc0 = wave_sum(p0.dot(sAa0, wb0));
      double acc1 = 0.0;
      if (TWO) { p1.opaque(); acc1 = wave_sum(p1.dot(sAa1, wb1)); }
      if (lane == 0) { red[parity][wv][0] = acc0; if (TWO) red[parity][wv][1] = acc1; }
      __syncthreads();
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { t0 += red[parity][w][0]; if (TWO) t1 += red[parity][w][1]; }
      parity ^= 1;
      double ti = t0 / rowcnt0 * (double)p0.P;                     // missingvals.py:37 (0 / 0 -> NaN for an empty row)
      if (TWO) ti = (ti + t1 / rowcnt1 * (double)p1.P) / 2.0;      // np.average(Ts, axis=0), cmtf.py:155,206
      // the reference keeps its mask from the INPUT: once a score is NaN the deflated copy is NaN at observed positions
      // too, and every later score of the sample is NaN (here those entries would otherwise read as missing)
      bad = bad || (ti != ti);
      if (bad) ti = __builtin_nan("");
      if (threadIdx.x == 0) scores[row * (int64_t)ld + a] = ti;
      p0.opaque();
      p0.deflate(ti, sAa0, wb0);
      if (TWO) { p1.opaque(); p1.deflate(ti, sAa1, wb1); }
    }
  }
}

constexpr size_t kProjLdsMax = 144 * 1024;

template <typename T>
static bool block_fits(const ProjBlock<T>& b, int NT, int max_nv, int* nv_out) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t P = (int64_t)b.A * b.B, stride = (int64_t)NT * V;
  const int64_t nv = (P + stride - 1) / stride;
  // B % V == 0 (a vector never straddles a j boundary); stride % B == 0 (k constant per lane)
  if ((b.B % V) != 0 || (stride % b.B) != 0 || nv > max_nv || (reinterpret_cast<uintptr_t>(b.X) & 15) != 0 || ((P * sizeof(T)) & 15) != 0) return false;
  *nv_out = (int)nv;
  return true;
}

template <typename T>
static int run_project_rows(ProjBlock<T> b0, ProjBlock<T> b1, int nblocks, int64_t I, int R, double* scores, int ld, hipStream_t st,
                            const int64_t* rows = nullptr) {
  if (nblocks < 1 || nblocks > 2 || !b0.X || !b0.WA || !b0.WB || !scores || I <= 0 || b0.A <= 0 || b0.B <= 0 || R <= 0 || ld < R ||
      (nblocks == 2 && (!b1.X || !b1.WA || !b1.WB || b1.A <= 0 || b1.B <= 0))) {
    set_error("project_rows: bad argument");
    return CMTFPLS_EINVAL;
  }
  const char* outside = "project_rows: shape outside the row-in-registers form; use the score_deflate passes";
  size_t lds = (size_t)R * (size_t)(b0.A + b0.B) * sizeof(double);
  if (nblocks == 2) lds += (size_t)R * (size_t)(b1.A + b1.B) * sizeof(double);
  if (lds > kProjLdsMax) { set_error(outside); return CMTFPLS_EUNSUPPORTED; }
  const dim3 g((unsigned)(I < 4096 ? I : 4096));
#define PRL(NTT, NVA, NVB)                                                                                               \
  do {                                                                                                                   \
    if (lds + 1024 > 64 * 1024)                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(project_rows_kernel<T, NTT, NVA, NVB>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((project_rows_kernel<T, NTT, NVA, NVB>), g, dim3(NTT), lds, st, b0, b1, I, R, scores, ld, rows);  \
  } while (0)
  int nv0 = 0, nv1 = 0;
  if (nblocks == 1) {
    if (block_fits(b0, 256, 16, &nv0)) {
      if (nv0 <= 2) PRL(256, 2, 0); else if (nv0 <= 4) PRL(256, 4, 0); else if (nv0 <= 8) PRL(256, 8, 0); else PRL(256, 16, 0);
    } else if (block_fits(b0, 1024, 16, &nv0)) {
      if (nv0 <= 8) PRL(1024, 8, 0); else PRL(1024, 16, 0);
    } else {
      set_error(outside);
      return CMTFPLS_EUNSUPPORTED;
    }
    return check_launch("project_rows");
  }
  // two coupled blocks: the longer one first (the score is their mean: the order does not matter)
  if ((int64_t)b1.A * b1.B > (int64_t)b0.A * b0.B) { const ProjBlock<T> t = b0; b0 = b1; b1 = t; }
  if (!block_fits(b0, 256, 16, &nv0) || !block_fits(b1, 256, 4, &nv1) || nv0 + nv1 > 17) { set_error(outside); return CMTFPLS_EUNSUPPORTED; }
  const int s0 = nv0 <= 2 ? 2 : nv0 <= 4 ? 4 : nv0 <= 8 ? 8 : 16, s1 = nv1 <= 1 ? 1 : 4;
  if (s0 == 16 && s1 == 4) { set_error(outside); return CMTFPLS_EUNSUPPORTED; }
  switch (s0 * 8 + s1) {
    case 2 * 8 + 1: PRL(256, 2, 1); break;
    case 2 * 8 + 4: PRL(256, 2, 4); break;
    case 4 * 8 + 1: PRL(256, 4, 1); break;
    case 4 * 8 + 4: PRL(256, 4, 4); break;
    case 8 * 8 + 1: PRL(256, 8, 1); break;
    case 8 * 8 + 4: PRL(256, 8, 4); break;
    default: PRL(256, 16, 1); break;
  }
#undef PRL
  return check_launch("project_rows");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
int cmtfpls_project_rows_f32(const float* X, int64_t I, int A, int B, int R, const double* WA, const double* WB, const double* mean,
                             double* scores, int ld, void* stream) {
  return run_project_rows<float>(ProjBlock<float>{X, WA, WB, mean, A, B}, ProjBlock<float>{nullptr, nullptr, nullptr, nullptr, 0, 0}, 1, I, R,
                                 scores, ld, (hipStream_t)stream);
}
int cmtfpls_project_rows_f64(const double* X, int64_t I, int A, int B, int R, const double* WA, const double* WB, const double* mean,
                             double* scores, int ld, void* stream) {
  return run_project_rows<double>(ProjBlock<double>{X, WA, WB, mean, A, B}, ProjBlock<double>{nullptr, nullptr, nullptr, nullptr, 0, 0}, 1, I, R,
                                  scores, ld, (hipStream_t)stream);
}
int cmtfpls_project_rows_idx_f32(const float* X, const int64_t* rows, int64_t n_rows, int A, int B, int R, const double* WA, const double* WB,
                                 const double* mean, double* scores, int ld, void* stream) {
  if (!rows) { set_error("project_rows_idx: bad argument"); return CMTFPLS_EINVAL; }
  if (n_rows == 0) return CMTFPLS_OK;
  return run_project_rows<float>(ProjBlock<float>{X, WA, WB, mean, A, B}, ProjBlock<float>{nullptr, nullptr, nullptr, nullptr, 0, 0}, 1, n_rows, R,
                                 scores, ld, (hipStream_t)stream, rows);
}
int cmtfpls_project_rows_idx_f64(const double* X, const int64_t* rows, int64_t n_rows, int A, int B, int R, const double* WA, const double* WB,
                                 const double* mean, double* scores, int ld, void* stream) {
  if (!rows) { set_error("project_rows_idx: bad argument"); return CMTFPLS_EINVAL; }
  if (n_rows == 0) return CMTFPLS_OK;
  return run_project_rows<double>(ProjBlock<double>{X, WA, WB, mean, A, B}, ProjBlock<double>{nullptr, nullptr, nullptr, nullptr, 0, 0}, 1, n_rows, R,
                                  scores, ld, (hipStream_t)stream, rows);
}
int cmtfpls_project_rows2_idx_f32(const float* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                                  const float* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                                  const int64_t* rows, int64_t n_rows, int R, double* scores, int ld, void* stream) {
  if (!rows) { set_error("project_rows2_idx: bad argument"); return CMTFPLS_EINVAL; }
  if (n_rows == 0) return CMTFPLS_OK;
  return run_project_rows<float>(ProjBlock<float>{X0, WA0, WB0, mean0, A0, B0}, ProjBlock<float>{X1, WA1, WB1, mean1, A1, B1}, 2, n_rows, R,
                                 scores, ld, (hipStream_t)stream, rows);
}
int cmtfpls_project_rows2_idx_f64(const double* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                                  const double* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                                  const int64_t* rows, int64_t n_rows, int R, double* scores, int ld, void* stream) {
  if (!rows) { set_error("project_rows2_idx: bad argument"); return CMTFPLS_EINVAL; }
  if (n_rows == 0) return CMTFPLS_OK;
  return run_project_rows<double>(ProjBlock<double>{X0, WA0, WB0, mean0, A0, B0}, ProjBlock<double>{X1, WA1, WB1, mean1, A1, B1}, 2, n_rows, R,
                                  scores, ld, (hipStream_t)stream, rows);
}
int cmtfpls_project_rows2_f32(const float* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                              const float* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                              int64_t I, int R, double* scores, int ld, void* stream) {
  return run_project_rows<float>(ProjBlock<float>{X0, WA0, WB0, mean0, A0, B0}, ProjBlock<float>{X1, WA1, WB1, mean1, A1, B1}, 2, I, R,
                                 scores, ld, (hipStream_t)stream);
}
int cmtfpls_project_rows2_f64(const double* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                              const double* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                              int64_t I, int R, double* scores, int ld, void* stream) {
  return run_project_rows<double>(ProjBlock<double>{X0, WA0, WB0, mean0, A0, B0}, ProjBlock<double>{X1, WA1, WB1, mean1, A1, B1}, 2, I, R,
                                  scores, ld, (hipStream_t)stream);
}
}
