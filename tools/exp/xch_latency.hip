// Cross-workgroup exchange latency on gfx950 without release fences: a value that is its own flag, written with ONE 8-byte
// agent-scope relaxed store and polled with agent-scope relaxed loads (the primitive of score_contract_split_kernel).
//   ping-pong   two workgroups (block 0 and block `other`) bounce a counter: one-way latency
//   barrier     W workgroups, each writes its step number into its own slot and polls all W slots (lanes of wavefront 0): time per
//               all-to-all step
//   counter     W workgroups, one fetch_add on a shared counter + poll until it reaches step * W
// Every spin is bounded (a stuck partner ends the kernel with a failure flag instead of hanging the GPU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/xch_latency.hip -o tools/exp/xch_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kSpin = 1 << 20;

__global__ void pingpong(unsigned long long* a, unsigned long long* b, int n, int other, int* fail) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    for (int i = 1; i <= n; ++i) {
      __hip_atomic_store(a, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int s = 0;
      while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i)
        if (++s > kSpin) { *fail = 1; return; }
    }
  } else if ((int)blockIdx.x == other) {
    for (int i = 1; i <= n; ++i) {
      int s = 0;
      while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i)
        if (++s > kSpin) { *fail = 1; return; }
      __hip_atomic_store(b, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// slots: W consecutive 8-byte words (pad = 1) or one per 128-byte line (pad = 16)
__global__ void slot_barrier(unsigned long long* slots, int W, int pad, int n, int* fail) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  for (int i = 1; i <= n; ++i) {
    if (lane == 0) __hip_atomic_store(slots + (size_t)blockIdx.x * pad, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int s = 0;
    for (;;) {
      bool ok = true;
      for (int w = lane; w < W; w += 64)
        ok = ok && __hip_atomic_load(slots + (size_t)w * pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)i;
      if (__all(ok)) break;
      if (++s > kSpin) { *fail = 1; return; }
    }
  }
}

__global__ void counter_barrier(unsigned long long* ctr, int W, int n, int* fail) {
  if (threadIdx.x != 0) return;
  for (int i = 1; i <= n; ++i) {
    __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int s = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)i * W)
      if (++s > kSpin) { *fail = 1; return; }
  }
}

// the same all-to-all step with a release / acquire pair around it (what a data exchange through ordinary stores needs)
__global__ void slot_barrier_fenced(unsigned long long* slots, double* data, int W, int n, int* fail) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  for (int i = 1; i <= n; ++i) {
    data[(size_t)blockIdx.x * 64 + lane] = (double)i;                      // ordinary stores the partners will read
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(slots + (size_t)blockIdx.x * 16, (unsigned long long)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int s = 0;
    for (;;) {
      bool ok = true;
      for (int w = lane; w < W; w += 64)
        ok = ok && __hip_atomic_load(slots + (size_t)w * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)i;
      if (__all(ok)) break;
      if (++s > kSpin) { *fail = 1; return; }
    }
  }
}

int main() {
  unsigned long long* buf;
  double* data;
  int* fail;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMalloc(&data, 1 << 20));
  CK(hipMalloc(&fail, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n = 2000;
  auto run = [&](const char* name, auto launch, double per) {
    int hf = 0;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipMemset(buf, 0, 1 << 20);
      hipMemset(fail, 0, 4);
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f us per step%s\n", name, best * 1e3 / per, hf ? "   (SPIN LIMIT HIT: result invalid)" : "");
    fflush(stdout);
    return 0;
  };
  for (int other : {1, 8, 9, 64}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ping-pong blocks 0 <-> %d (one way)", other);
    run(nm, [&] { hipLaunchKernelGGL(pingpong, dim3(other + 1), dim3(64), 0, 0, buf, buf + 64, n, other, fail); }, 2.0 * n);
  }
  for (int W : {8, 64, 256}) {
    char nm[96];
    snprintf(nm, sizeof nm, "slot barrier, %d workgroups, packed slots", W);
    run(nm, [&] { hipLaunchKernelGGL(slot_barrier, dim3(W), dim3(64), 0, 0, buf, W, 1, n, fail); }, n);
    snprintf(nm, sizeof nm, "slot barrier, %d workgroups, a line per slot", W);
    run(nm, [&] { hipLaunchKernelGGL(slot_barrier, dim3(W), dim3(64), 0, 0, buf, W, 16, n, fail); }, n);
    snprintf(nm, sizeof nm, "counter barrier, %d workgroups", W);
    run(nm, [&] { hipLaunchKernelGGL(counter_barrier, dim3(W), dim3(64), 0, 0, buf, W, n, fail); }, n);
    snprintf(nm, sizeof nm, "slot barrier + release/acquire, %d workgroups", W);
    run(nm, [&] { hipLaunchKernelGGL(slot_barrier_fenced, dim3(W), dim3(64), 0, 0, buf, data, W, n, fail); }, n);
  }
  // reference: an empty kernel launched back to back (the launch boundary a barrier would replace)
  {
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(counter_barrier, dim3(64), dim3(64), 0, 0, buf, 64, 0, fail);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f us per launch\n", "empty kernel, 64 workgroups, back to back", ms * 1e3 / 200);
  }
  return 0;
}
