// Experiment (not part of the library): does the 256 MiB Infinity Cache speed up a streaming re-read of X?
//   (1) back-to-back forward sweeps over a buffer of S bytes, non-temporal vs default-policy 16-byte loads;
//   (2) "serpentine": forward sweep then backward sweep over the same S bytes, so that the tail of one sweep is the
//       head of the next (what a contraction sweep followed by a reversed score sweep would see).
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/mall_exp.hip -o tools/exp/mall_exp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kThreads = 256;
constexpr int kUnroll = 4;

template <bool NT>
__device__ __forceinline__ f4 ld(const f4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// chunk = kThreads * kUnroll float4 (16 KB); block b handles chunks b, b + G, ...; REV walks the chunks from the end.
template <bool NT, bool REV>
__global__ __launch_bounds__(kThreads) void sweep(const f4* __restrict__ src, int64_t nchunks, float* __restrict__ sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t cc = REV ? nchunks - 1 - c : c;
    const f4* p = src + cc * (kThreads * kUnroll) + threadIdx.x;
    f4 v[kUnroll];
#pragma unroll
    for (int s = 0; s < kUnroll; ++s) v[s] = ld<NT>(p + s * kThreads);
#pragma unroll
    for (int s = 0; s < kUnroll; ++s) acc += v[s];
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[0] = acc.x;   // never true: keeps the loads
}

template <bool NT, bool REV>
static float timed(const f4* buf, int64_t bytes, int blocks, float* sink, hipEvent_t a, hipEvent_t b) {
  const int64_t nchunks = bytes / (int64_t)(kThreads * kUnroll * 16);
  CHECK(hipEventRecord(a, 0));
  sweep<NT, REV><<<blocks, kThreads, 0, 0>>>(buf, nchunks, sink);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  const int64_t total = (int64_t)4608 << 20;
  f4* buf = nullptr;
  float* sink = nullptr;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, total));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int reps = 12;
  const int64_t sizes_mb[] = {64, 128, 192, 224, 256, 384, 512, 1024, 4096};
  for (int blocks : {2048, 4096}) {
    std::printf("== %d blocks of %d threads, %d x 16 B per thread per chunk\n", blocks, kThreads, kUnroll);
    std::printf("%8s | %28s | %28s | %28s | %28s\n", "MiB", "fwd,fwd nt  (us, TB/s)", "fwd,fwd default", "fwd,rev nt", "fwd,rev default");
    for (int64_t mb : sizes_mb) {
      const int64_t bytes = mb << 20;
      double t[4] = {0, 0, 0, 0};
      for (int mode = 0; mode < 4; ++mode) {
        // flush: stream the last 512 MiB of the big buffer so every size starts from the same cache state
        timed<true, false>(buf + ((total - ((int64_t)512 << 20)) >> 4), (int64_t)512 << 20, blocks, sink, a, b);
        double sum = 0;
        int cnt = 0;
        for (int r = 0; r < reps; ++r) {
          float ms;
          const bool rev = (mode >= 2) && (r & 1);
          if (mode == 0 || mode == 2) ms = rev ? timed<true, true>(buf, bytes, blocks, sink, a, b) : timed<true, false>(buf, bytes, blocks, sink, a, b);
          else ms = rev ? timed<false, true>(buf, bytes, blocks, sink, a, b) : timed<false, false>(buf, bytes, blocks, sink, a, b);
          if (r >= 4) { sum += ms; ++cnt; }
        }
        t[mode] = sum / cnt;
      }
      std::printf("%8lld |", (long long)mb);
      for (int mode = 0; mode < 4; ++mode) std::printf(" %12.1f us %8.2f TB/s |", t[mode] * 1e3, bytes / (t[mode] * 1e-3) / 1e12);
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  return 0;
}
