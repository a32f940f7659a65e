// Measured HBM ceilings for the roofline denominators ("nominal 8 TB/s, re-measured with a
// device copy kernel on the box").  Plain 16-byte-per-lane non-temporal streaming kernels with the access
// patterns of the sweeps and nothing else: read-only (the contraction / score pattern), read-modify-write
// in place (the deflation / centring pattern) and read -> write into a second buffer.  bench.py times
// them on an X-sized buffer in the same run and reports the sweeps against these, next to the 8 TB/s spec.
//
// Two workgroup -> address maps: flat (row_bytes == 0: lane l of workgroup b starts at vector b*256 + l and
// strides by the whole grid: consecutive workgroups are adjacent) and chunked (row_bytes > 0: a workgroup
// owns row_bytes contiguous bytes at a time and grid-strides over such rows -- the map of the row-wise
// sweeps, where a row of X is one chunk).
#include "common.hpp"

namespace cmtfpls {

constexpr int kCeilUnroll = 8;   // 16-byte accesses in flight per lane

// OP 0: read (sum kept alive), 1: in-place negate (an involution: two launches restore the buffer), 2: copy
template <int OP>
__device__ __forceinline__ void ceiling_span(const nt_f4* __restrict__ src, nt_f4* __restrict__ dst, int64_t first,
                                             int64_t n, int64_t stride, float& acc) {
  int64_t i = first;
  for (; i + (kCeilUnroll - 1) * stride < n; i += kCeilUnroll * stride) {
    nt_f4 v[kCeilUnroll];
#pragma unroll
    for (int s = 0; s < kCeilUnroll; ++s) v[s] = __builtin_nontemporal_load(src + i + s * stride);
#pragma unroll
    for (int s = 0; s < kCeilUnroll; ++s) {
      if (OP == 0) acc += (v[s].x + v[s].y) + (v[s].z + v[s].w);
      if (OP == 1) __builtin_nontemporal_store(-v[s], dst + i + s * stride);
      if (OP == 2) __builtin_nontemporal_store(v[s], dst + i + s * stride);
    }
  }
  for (; i < n; i += stride) {
    const nt_f4 v = __builtin_nontemporal_load(src + i);
    if (OP == 0) acc += (v.x + v.y) + (v.z + v.w);
    if (OP == 1) __builtin_nontemporal_store(-v, dst + i);
    if (OP == 2) __builtin_nontemporal_store(v, dst + i);
  }
}

template <int OP>
__global__ __launch_bounds__(kSweepThreads) void ceiling_kernel(const nt_f4* __restrict__ src, nt_f4* __restrict__ dst,
                                                               int64_t nvec, int64_t rowvec, float* __restrict__ sink) {
  __shared__ double red[16];
  float acc = 0.f;
  if (rowvec == 0) {
    ceiling_span<OP>(src, dst, (int64_t)blockIdx.x * kSweepThreads + threadIdx.x, nvec, (int64_t)gridDim.x * kSweepThreads, acc);
  } else {
    const int64_t nrows = nvec / rowvec;           // a ragged tail (< one row) is not swept: sizes are reported from nrows
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x)
      ceiling_span<OP>(src + r * rowvec, dst + r * rowvec, threadIdx.x, rowvec, kSweepThreads, acc);
  }
  if (OP == 0) {
    const double s = block_sum((double)acc, red);
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)s;
  }
}

static int ceiling_args_ok(const void* a, const void* b, size_t bytes, int blocks, int64_t row_bytes) {
  return a && b && bytes >= 16 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && blocks > 0 &&
         blocks <= 4096 && row_bytes >= 0 && (row_bytes % 16) == 0 && (row_bytes == 0 || (size_t)row_bytes <= bytes);
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

int cmtfpls_ceiling_max_blocks(void) { return 4096; }

int cmtfpls_ceiling_read(const void* buf, size_t bytes, int64_t row_bytes, float* sink, int blocks, void* stream) {
  if (!ceiling_args_ok(buf, sink, bytes, blocks, row_bytes)) { set_error("ceiling_read: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL((ceiling_kernel<0>), dim3(blocks), dim3(kSweepThreads), 0, (hipStream_t)stream,
                     static_cast<const nt_f4*>(buf), (nt_f4*)nullptr, (int64_t)(bytes / 16), row_bytes / 16, sink);
  return check_launch("ceiling_read");
}

int cmtfpls_ceiling_rmw(void* buf, size_t bytes, int64_t row_bytes, int blocks, void* stream) {
  if (!ceiling_args_ok(buf, buf, bytes, blocks, row_bytes)) { set_error("ceiling_rmw: bad argument"); return CMTFPLS_EINVAL; }
  nt_f4* p = static_cast<nt_f4*>(buf);
  hipLaunchKernelGGL((ceiling_kernel<1>), dim3(blocks), dim3(kSweepThreads), 0, (hipStream_t)stream, p, p, (int64_t)(bytes / 16),
                     row_bytes / 16, (float*)nullptr);
  return check_launch("ceiling_rmw");
}

int cmtfpls_ceiling_copy(const void* src, void* dst, size_t bytes, int64_t row_bytes, int blocks, void* stream) {
  if (!ceiling_args_ok(src, dst, bytes, blocks, row_bytes) || src == dst) { set_error("ceiling_copy: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL((ceiling_kernel<2>), dim3(blocks), dim3(kSweepThreads), 0, (hipStream_t)stream,
                     static_cast<const nt_f4*>(src), static_cast<nt_f4*>(dst), (int64_t)(bytes / 16), row_bytes / 16, (float*)nullptr);
  return check_launch("ceiling_copy");
}

}  // extern "C"
