// Experiment (not part of the library): cycles per f64 MFMA on gfx950, back-to-back issue from registers, one and two
// waves per SIMD, for the two f64 shapes -- v_mfma_f64_16x16x4_f64 (2048 FMA-flops... 16*16*4*2 = 2048 flop) and
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: 512 flop).  Question (round 3, MTTKRP with R = 10): is the 4x4x4 form the
// same flop rate as the 16x16x4 one?  If so, three 4-wide component groups (12 columns) cost 25 % fewer matrix cycles
// than one 16-wide tile.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_f64_rate.hip -o tools/exp/mfma_f64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// NACC independent accumulators, N iterations of NACC back-to-back MFMAs each
template <int NACC>
__global__ __launch_bounds__(512) void k16(double a, double b, int n, double* out, long long* cyc) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  const double av = a + threadIdx.x * 1e-9, bv = b - threadIdx.x * 1e-9;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(512) void k4(double a, double b, int n, double* out, long long* cyc) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  const double av = a + threadIdx.x * 1e-9, bv = b - threadIdx.x * 1e-9;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int grid, int threads, int n, int nacc, double flop_per) {
  double* out;
  long long* cyc;
  CHECK(hipMalloc(&out, sizeof(double) * grid * threads));
  CHECK(hipMalloc(&cyc, sizeof(long long) * grid));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch(grid, threads, n, out, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch(grid, threads, n, out, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long c0;
  CHECK(hipMemcpy(&c0, cyc, sizeof(long long), hipMemcpyDeviceToHost));
  const double waves = (double)grid * threads / 64.0;
  const double mfma_per_wave = (double)n * nacc;
  // s_memtime ticks at 100 MHz on gfx9?  report both: wall-derived and the stamp
  std::printf("%-34s grid %5d x %4d thr, %d acc: %8.3f ms  %7.2f TFLOP/s  wall ns per MFMA per wave-slot %.2f  memtime ticks/MFMA %.2f\n", name, grid,
              threads, nacc, ms, waves * mfma_per_wave * flop_per / ms / 1e9, ms * 1e6 / mfma_per_wave, (double)c0 / mfma_per_wave);
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
}

int main() {
  const int n = 20000;
  for (int threads : {256, 512}) {             // 256 threads = 4 waves = one per SIMD; 512 = two per SIMD
    for (int grid : {1, 256, 512}) {
      run("v_mfma_f64_16x16x4_f64 1 acc", [](int g, int t, int n, double* o, long long* c) { hipLaunchKernelGGL(k16<1>, dim3(g), dim3(t), 0, 0, 1.0, 2.0, n, o, c); }, grid, threads, n, 1, 2048.0);
      run("v_mfma_f64_16x16x4_f64 4 acc", [](int g, int t, int n, double* o, long long* c) { hipLaunchKernelGGL(k16<4>, dim3(g), dim3(t), 0, 0, 1.0, 2.0, n, o, c); }, grid, threads, n, 4, 2048.0);
      run("v_mfma_f64_4x4x4_4b_f64 1 acc", [](int g, int t, int n, double* o, long long* c) { hipLaunchKernelGGL(k4<1>, dim3(g), dim3(t), 0, 0, 1.0, 2.0, n, o, c); }, grid, threads, n, 1, 512.0);
      run("v_mfma_f64_4x4x4_4b_f64 3 acc", [](int g, int t, int n, double* o, long long* c) { hipLaunchKernelGGL(k4<3>, dim3(g), dim3(t), 0, 0, 1.0, 2.0, n, o, c); }, grid, threads, n, 3, 512.0);
      run("v_mfma_f64_4x4x4_4b_f64 12 acc", [](int g, int t, int n, double* o, long long* c) { hipLaunchKernelGGL(k4<12>, dim3(g), dim3(t), 0, 0, 1.0, 2.0, n, o, c); }, grid, threads, n, 12, 512.0);
    }
  }
  return 0;
}
