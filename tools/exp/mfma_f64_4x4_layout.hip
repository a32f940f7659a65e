// Experiment: operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by probing: for every (la, lb) put 1.0 in lane
// la of A and lane lb of B (zeros elsewhere) and record which lane of D becomes 1.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_f64_4x4_layout.hip -o tools/exp/mfma_f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(64) void probe(int* table) {   // table[la * 64 + lb] = lane of D that received the product, or -1
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) table[la * 64 + lb] = l;
    }
}

int main() {
  int* t;
  CHECK(hipMalloc(&t, 4096 * sizeof(int)));
  CHECK(hipMemset(t, 0xff, 4096 * sizeof(int)));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t);
  CHECK(hipDeviceSynchronize());
  static int h[4096];
  CHECK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
  // for every A lane: which B lanes pair with it, and where the product lands
  for (int la = 0; la < 64; ++la) {
    std::printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) std::printf("  B%-2d->D%-2d", lb, h[la * 64 + lb]);
    std::printf("\n");
  }
  return 0;
}
