// transform / predict for inputs WITH missing values (tpls.py:128-142, 151-165 with miss_mmodedot, missingvals.py:23-38):
// the reference centres a copy of X and then, R times, projects every sample on the component's loadings (per-row rescale
// P / n_observed) and deflates the copy -- R read + write passes (cmtfpls_score_deflate_*).  The samples are independent,
// so here a 1024-thread workgroup takes one ROW, keeps it in registers, and runs the whole sequence on it: centring
// (x - mean, rounded to the storage type as the stored copy would be), the observation count, and for a = 0..R-1 the
// masked score, the rescale and the rank-one deflation (rounded to the storage type again, as the write-back would).
// X is read ONCE and never written; the arithmetic per element and the summation order (per lane over its vectors, butterfly
// wave sum, wavefronts in index order) are those of the sequential kernels.  The loadings of all R components sit in LDS,
// component-major.  Rows of up to 256 * V * 16 elements (128 x 128 f32, 64 x 128 f64) whose trailing extent B divides the
// workgroup stride (256 * V elements: every power-of-two B up to 1024 f32 / 512 f64); other shapes keep the passes.
#include "common.hpp"

namespace cmtfpls {

constexpr int kProjThreads = 256;

// NV vectors of 16 bytes per lane (the row is NV * 256 vectors long at most).  The workgroup stride (256 * V elements) is a
// multiple of B ("k constant", as in score_deflate_kernel): every vector a lane owns has the same k = c % B, so the V
// entries of wB it needs are read from LDS once per component, and j advances by the uniform dj = stride / B.
// 256-thread workgroups: four rows in flight per CU, and the barrier of a step spans four wavefronts, not sixteen.
template <typename T, int NV>
__global__ __launch_bounds__(kProjThreads) void project_rows_kernel(const T* __restrict__ X, int64_t I, int A, int B, int R,
                                                                   const double* __restrict__ WA, const double* __restrict__ WB,
                                                                   const double* __restrict__ mean, double* __restrict__ scores,
                                                                   int ld) {
  extern __shared__ double lds[];                    // sA[R][A] | sB[R][B]
  __shared__ double red[2][kProjThreads / 64];
  constexpr int V = VecOf<T>::N;
  using VT = Pack<T, V>;
  double* sA = lds;
  double* sB = lds + (size_t)R * A;
  for (int idx = threadIdx.x; idx < R * A; idx += kProjThreads) { const int a = idx / A, j = idx % A; sA[idx] = WA[(int64_t)j * R + a]; }
  for (int idx = threadIdx.x; idx < R * B; idx += kProjThreads) { const int a = idx / B, k = idx % B; sB[idx] = WB[(int64_t)k * R + a]; }
  const unsigned P = (unsigned)A * (unsigned)B;
  constexpr unsigned stride = kProjThreads * V;
  const unsigned c0 = threadIdx.x * V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = (int)(c0 / (unsigned)B), k0 = (int)(c0 % (unsigned)B), dj = (int)(stride / (unsigned)B);
  __syncthreads();
  int parity = 0;
  for (int64_t row = blockIdx.x; row < I; row += gridDim.x) {
    // scalar row base (opaque to loop strength reduction, which otherwise keeps one 64-bit running address per vector in
    // VGPRs): every access is SGPR base + one shared VGPR offset
    const uint64_t rb = reinterpret_cast<uint64_t>(X + row * (int64_t)P);
    const uint32_t rb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rb >> 32));
    const uint32_t rb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb);
    const T* __restrict__ xr = reinterpret_cast<const T*>(((uint64_t)rb_hi << 32) | (uint64_t)rb_lo);
    VT x[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (c0 + n * stride < P) x[n] = ld_stream(reinterpret_cast<const VT*>((xr + (int64_t)n * stride) + c0));
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
    cnt = wave_sum(cnt);
    if (lane == 0) red[parity][wv] = cnt;
    __syncthreads();
    double rowcnt = 0.0;
#pragma unroll
    for (int w = 0; w < kProjThreads / 64; ++w) rowcnt += red[parity][w];
    parity ^= 1;
#pragma unroll 1
    for (int a = 0; a < R; ++a) {
      const double* __restrict__ sAa = sA + (size_t)a * A;
      double wb[V];
#pragma unroll
      for (int e = 0; e < V; ++e) wb[e] = sB[(size_t)a * B + k0 + e];
      // keep the row opaque between the phases: otherwise its f64 conversions are kept alive for reuse (2 VGPRs per element)
#pragma unroll
      for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < V; ++e) asm volatile("" : "+v"(x[n].e[e]));
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
          asm volatile("" : "+v"(jn));                             // no per-vector LDS address kept (and advanced) across the loops
          acc = fma(sAa[jn], d, acc);
          if (NV > 4) __builtin_amdgcn_sched_barrier(0);          // keep live temporaries low: the row owns the VGPRs
        }
      acc = wave_sum(acc);
      if (lane == 0) red[parity][wv] = acc;
      __syncthreads();
      double ti = 0.0;
#pragma unroll
      for (int w = 0; w < kProjThreads / 64; ++w) ti += red[parity][w];
      parity ^= 1;
      ti = ti / rowcnt * (double)P;                                 // missingvals.py:37 (0 / 0 -> NaN for an empty row)
      if (threadIdx.x == 0) scores[row * (int64_t)ld + a] = ti;
#pragma unroll
      for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < V; ++e) asm volatile("" : "+v"(x[n].e[e]));
#pragma unroll
      for (int n = 0; n < NV; ++n)
        if (c0 + n * stride < P) {
          int jn = j0 + n * dj;
          asm volatile("" : "+v"(jn));
          const double tw = ti * sAa[jn];
#pragma unroll
          for (int e = 0; e < V; ++e) x[n].e[e] = (T)fma(-tw, wb[e], (double)x[n].e[e]);   // tpls.py:142 on the stored type
          if (NV > 4) __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
}

template <typename T>
static int run_project_rows(const T* X, int64_t I, int A, int B, int R, const double* WA, const double* WB, const double* mean,
                            double* scores, int ld, hipStream_t st) {
  if (!X || !WA || !WB || !scores || I <= 0 || A <= 0 || B <= 0 || R <= 0 || ld < R) { set_error("project_rows: bad argument"); return CMTFPLS_EINVAL; }
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t P = (int64_t)A * B;
  const size_t lds = (size_t)R * (size_t)(A + B) * sizeof(double);
  const int64_t stride = (int64_t)kProjThreads * V;
  const int64_t nv = (P + stride - 1) / stride;
  // B % V == 0 (a vector never straddles a j boundary); stride % B == 0 (k constant per lane); 16 vectors per lane at most
  // (LDS: the loadings plus the 64 static bytes of `red`; beyond 64 KB the dynamic limit is raised per instance below)
  constexpr size_t kProjLdsMax = 144 * 1024;
  if ((B % V) != 0 || (stride % B) != 0 || nv > 16 || lds > kProjLdsMax || (reinterpret_cast<uintptr_t>(X) & 15) != 0) {
    set_error("project_rows: shape outside the row-in-registers form; use the score_deflate passes");
    return CMTFPLS_EUNSUPPORTED;
  }
  const int grid = (int)(I < 4096 ? I : 4096);
  const dim3 g(grid), b(kProjThreads);
#define PRL(NVV)                                                                                                         \
  do {                                                                                                                   \
    if (lds + 64 > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(project_rows_kernel<T, NVV>),                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((project_rows_kernel<T, NVV>), g, b, lds, st, X, I, A, B, R, WA, WB, mean, scores, ld);           \
  } while (0)
  if (nv <= 2) PRL(2); else if (nv <= 4) PRL(4); else if (nv <= 8) PRL(8); else PRL(16);
#undef PRL
  return check_launch("project_rows");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
int cmtfpls_project_rows_f32(const float* X, int64_t I, int A, int B, int R, const double* WA, const double* WB, const double* mean,
                             double* scores, int ld, void* stream) {
  return run_project_rows<float>(X, I, A, B, R, WA, WB, mean, scores, ld, (hipStream_t)stream);
}
int cmtfpls_project_rows_f64(const double* X, int64_t I, int A, int B, int R, const double* WA, const double* WB, const double* mean,
                             double* scores, int ld, void* stream) {
  return run_project_rows<double>(X, I, A, B, R, WA, WB, mean, scores, ld, (hipStream_t)stream);
}
}
