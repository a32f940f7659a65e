// Experiment: which workgroup -> address map gives the mode-0 contraction (Z[c] = sum_i X[i,c] u[i], f32 X, f64
// accumulators) the most bandwidth?  u precomputed; partial rows written, not reduced (the reduce is a second kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// (a) column owner: 256 threads x U vectors, RU rows in flight, grid (col tiles, row blocks)   [the product kernel's map]
template <int U, int RU>
__global__ __launch_bounds__(256) void colowner_kernel(const float* __restrict__ X, int64_t I, int64_t P, const double* __restrict__ u,
                                                      double* __restrict__ part, int rows_per_block) {
  const int64_t cbase = (int64_t)blockIdx.x * (256 * 4 * U) + (int64_t)threadIdx.x * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < I ? r0 + rows_per_block : I;
  double acc[U][4];
#pragma unroll
  for (int g = 0; g < U; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[g][e] = 0.0;
  for (int64_t r = r0; r + RU <= r1; r += RU) {
    f4 x[RU][U];
    double uu[RU];
#pragma unroll
    for (int s = 0; s < RU; ++s) {
      uu[s] = u[r + s];
#pragma unroll
      for (int g = 0; g < U; ++g) x[s][g] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(X + (r + s) * P + cbase + (int64_t)g * 1024));
    }
#pragma unroll
    for (int s = 0; s < RU; ++s)
#pragma unroll
      for (int g = 0; g < U; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g][e] = fma((double)x[s][g][e], uu[s], acc[g][e]);
  }
#pragma unroll
  for (int g = 0; g < U; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(int64_t)blockIdx.y * P + cbase + (int64_t)g * 1024 + e] = acc[g][e];
}

// (b) row segment per 1024-thread workgroup: a lane owns NV vectors of its segment in every row the workgroup visits
// (segment = 1024 * 4 * NV columns; nseg segments per row), RU rows in flight
template <int NV, int RU>
__global__ __launch_bounds__(1024) void rowseg_kernel(const float* __restrict__ X, int64_t I, int64_t P, int nseg, const double* __restrict__ u,
                                                     double* __restrict__ part) {
  const int seg = blockIdx.x % nseg;
  const int64_t step = gridDim.x / nseg;
  const float* __restrict__ xs = X + (int64_t)seg * (1024 * 4 * NV) + (int64_t)threadIdx.x * 4;
  double acc[NV][4];
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[n][e] = 0.0;
  int64_t r = blockIdx.x / nseg;
  for (; r + (RU - 1) * step < I; r += RU * step) {
    f4 x[RU][NV];
    double uu[RU];
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      uu[q] = u[r + q * step];
#pragma unroll
      for (int n = 0; n < NV; ++n) x[q][n] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(xs + (r + q * step) * P + (int64_t)n * 4096));
    }
#pragma unroll
    for (int q = 0; q < RU; ++q)
#pragma unroll
      for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] = fma((double)x[q][n][e], uu[q], acc[n][e]);
  }
  double* __restrict__ prow = part + (int64_t)(blockIdx.x / nseg) * P + (int64_t)seg * (1024 * 4 * NV) + (int64_t)threadIdx.x * 4;
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < 4; ++e) prow[(int64_t)n * 4096 + e] = acc[n][e];
}

extern "C" int contract_exp(int kind, const float* X, int64_t I, int64_t P, const double* u, double* part, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
#define CO(ID, U, RU) case ID: { const int ct = (int)(P / (1024 * U)); const int rb = grid / ct; const int rpb = (int)((I + rb - 1) / rb); \
    hipLaunchKernelGGL((colowner_kernel<U, RU>), dim3(ct, rb), dim3(256), 0, st, X, I, P, u, part, rpb); } break;
    CO(0, 4, 4) CO(1, 2, 4) CO(2, 4, 2) CO(3, 2, 8)
#define RS(ID, NV, RU) case ID: { const int nseg = (int)(P / (4096 * NV)); hipLaunchKernelGGL((rowseg_kernel<NV, RU>), dim3(grid), dim3(1024), 0, st, X, I, P, nseg, u, part); } break;
    RS(10, 4, 2) RS(11, 4, 4) RS(12, 2, 2) RS(13, 2, 4) RS(14, 2, 8) RS(15, 1, 4) RS(16, 1, 8)
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
