// Experiment: what bounds the MTTKRP kernel?  Copies of its FAST path with parts switched off.
//   WMODE 0: B operand = constant (no LDS weight reads, no f64 multiply)      1: as the product (2 LDS reads + mul)
//   CMODE 0: A operand = raw register reinterpret (no f32->f64 convert)        1: convert
//   G: row groups per wavefront that share one B operand (register blocking; the product kernel has G = 1)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WMODE, int CMODE, int G, int UN>
__global__ __launch_bounds__(256) void mttkrp_exp_kernel(const float* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ WA,
                                                        const double* __restrict__ WB, int R, double* __restrict__ out, int ldo) {
  extern __shared__ double lds[];
  constexpr int RP = 16;
  double* sA = lds;
  double* sB = lds + (size_t)A * RP;
  for (int idx = threadIdx.x; idx < A * RP; idx += 256) { const int j = idx / RP, r = idx % RP; sA[idx] = (r < R) ? WA[(int64_t)j * R + r] : 0.0; }
  for (int idx = threadIdx.x; idx < B * RP; idx += 256) { const int k = idx / RP, r = idx % RP; sB[idx] = (r < R) ? WB[(int64_t)k * R + r] : 0.0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ri = lane & 15, kq = lane >> 4;
  const int64_t P = (int64_t)A * B;
  const int64_t nsuper = I / (16 * G);
  for (int64_t sg = (int64_t)blockIdx.x * 4 + wv; sg < nsuper; sg += (int64_t)gridDim.x * 4) {
    const float* __restrict__ xr = X + (sg * 16 * G + ri) * P;
    d4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = d4{0.0, 0.0, 0.0, 0.0};
    int j = (4 * kq) / B, k = (4 * kq) % B;
    for (int64_t c0 = 0; c0 < P; c0 += 16 * UN) {
      f4 x[UN][G];
#pragma unroll
      for (int s = 0; s < UN; ++s)
#pragma unroll
        for (int g = 0; g < G; ++g)
          x[s][g] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(xr + (int64_t)g * 16 * P + c0 + 16 * s + 4 * kq));
#pragma unroll
      for (int s = 0; s < UN; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double b;
          if (WMODE == 0) b = 1.0 + (double)(ri + e);
          else b = sA[(size_t)j * RP + ri] * sB[(size_t)(k + e) * RP + ri];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            double a;
            if (CMODE == 0) a = __longlong_as_double(((long long)__float_as_int(x[s][g][e]) << 29) + 0x3800000000000000LL);
            else a = (double)x[s][g][e];
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[g], 0, 0, 0);
          }
        }
        k += 16;
        if (k >= B) { k -= B; ++j; }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t row = sg * 16 * G + g * 16 + kq + 4 * q;
        if (ri < R) out[row * ldo + ri] = acc[g][q];
      }
  }
}

extern "C" int mttkrp_exp(int kind, const float* X, int64_t I, int A, int B, const double* WA, const double* WB, int R, double* out, int ldo, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(A + B) * 16 * sizeof(double);
#define L(ID, W, C, G, U) case ID: hipLaunchKernelGGL((mttkrp_exp_kernel<W, C, G, U>), dim3(grid), dim3(256), lds, st, X, I, A, B, WA, WB, R, out, ldo); break;
  switch (kind) {
    L(0, 1, 1, 1, 4)     // the product kernel's structure
    L(1, 0, 1, 1, 4)     // no weights
    L(2, 1, 0, 1, 4)     // no convert
    L(3, 0, 0, 1, 4)     // neither: loads + MFMA only
    L(4, 1, 1, 2, 4)     // B operand shared by 2 row groups
    L(5, 1, 1, 4, 2)     // ... by 4 row groups
    L(6, 1, 1, 4, 4)
    L(7, 1, 1, 2, 8)
    L(8, 0, 0, 4, 2)
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
