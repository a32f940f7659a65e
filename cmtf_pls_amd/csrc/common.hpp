// Shared device helpers for the gfx950 NIPALS kernels (64-wide wavefronts throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cmtfpls.h"

namespace cmtfpls {

constexpr int kWave = 64;
// Every X sweep (center / score / deflate / score_deflate) is launched with exactly this many
// workgroups, grid-striding over rows; 2048 x 256 threads = 8 workgroups on each of the 256 CUs.
constexpr int kSweepBlocks = 2048;
constexpr int kSweepThreads = 256;

// 16-byte vector of T: float4 / double2 loads and stores (global_load_dwordx4).
template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
  T e[N];
};
template <typename T>
struct VecOf {
  static constexpr int N = 16 / sizeof(T);
  using type = Pack<T, N>;
};

// Butterfly sum over the 64 lanes of a wavefront; every lane ends with the same bits
// (a + b == b + a exactly, and lanes l and l^m add the same two operands at every level).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// Sum over a workgroup of up to 16 wavefronts; all threads return the same value.
// `slot` is an LDS array of >= 16 doubles that the caller does not touch concurrently.
__device__ __forceinline__ double block_sum(double v, double* slot) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (lane == 0) slot[wv] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += slot[w];
  return s;
}

void set_error(const char* msg);
int check_launch(const char* what);

}  // namespace cmtfpls
