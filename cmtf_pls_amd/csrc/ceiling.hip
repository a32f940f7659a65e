// Measured HBM ceilings for the roofline denominators ("nominal 8 TB/s, re-measured with a
// device copy kernel on the box").  Plain 16-byte-per-lane non-temporal streaming kernels with the access
// patterns of the sweeps and nothing else: read-only (the contraction / score pattern), read-modify-write
// in place (the deflation / centring pattern) and read -> write into a second buffer.  bench.py times
// them on an X-sized buffer in the same run and reports the sweeps against these, next to the 8 TB/s spec.
//
// Four workgroup -> address maps (the access-pattern study behind them: tools/exp/rowexp.hip,
// profiles/r02b_access_pattern_experiments.txt): 0 flat (lane l of workgroup b starts at vector b*256 + l and
// strides by the whole grid), 1 chunked (a 256-thread workgroup owns row_bytes contiguous bytes at a time and
// grid-strides over such rows), 2 one 1024-thread workgroup per row with a barrier between its read burst and its
// write burst (the map of deflate_rows / center_rows / score_deflate), 3 column owner (a workgroup owns 8 KB of
// columns and a block of rows, 4 rows in flight: the map of the contraction and of deflate_contract).
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

// MAP 2: one 1024-thread workgroup per row (row_bytes = 1024 * 16 * NV): every lane loads its NV vectors, a barrier
// separates the read burst from the write burst (OP 1 / 2), then stores: the map of deflate_rows / center_rows /
// score_deflate.
template <int OP, int NV>
__global__ __launch_bounds__(1024) void ceiling_rowwg_kernel(const nt_f4* __restrict__ src, nt_f4* __restrict__ dst, int64_t nrows,
                                                            float* __restrict__ sink) {
  __shared__ double red[16];
  float acc = 0.f;
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const nt_f4* __restrict__ row = src + r * (int64_t)(1024 * NV);
    nt_f4 v[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) v[n] = __builtin_nontemporal_load(row + threadIdx.x + n * 1024);
    if (OP != 0) __syncthreads();
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (OP == 0) acc += (v[n].x + v[n].y) + (v[n].z + v[n].w);
      if (OP == 1) __builtin_nontemporal_store(-v[n], dst + r * (int64_t)(1024 * NV) + threadIdx.x + n * 1024);
      if (OP == 2) __builtin_nontemporal_store(v[n], dst + r * (int64_t)(1024 * NV) + threadIdx.x + n * 1024);
    }
  }
  if (OP == 0) {
    const double s = block_sum((double)acc, red);
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)s;
  }
}

// MAP 3: column owner (the contraction's map): a 256-thread workgroup owns 2 x 256 vectors of columns and a block of
// rows, 4 rows in flight (8 loads per lane), rows of rowvec vectors; grid = (rowvec / 512, row blocks).
template <int OP>
__global__ __launch_bounds__(kSweepThreads) void ceiling_colowner_kernel(const nt_f4* __restrict__ src, nt_f4* __restrict__ dst,
                                                                        int64_t nrows, int64_t rowvec, int rows_per_block,
                                                                        float* __restrict__ sink) {
  __shared__ double red[16];
  constexpr int NG = 2, RU = 4;
  const int64_t c = (int64_t)blockIdx.x * kSweepThreads * NG + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < nrows ? r0 + rows_per_block : nrows;
  float acc = 0.f;
  for (int64_t r = r0; r + RU <= r1; r += RU) {
    nt_f4 v[RU][NG];
#pragma unroll
    for (int s = 0; s < RU; ++s)
#pragma unroll
      for (int g = 0; g < NG; ++g) v[s][g] = __builtin_nontemporal_load(src + (r + s) * rowvec + c + g * kSweepThreads);
#pragma unroll
    for (int s = 0; s < RU; ++s)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (OP == 0) acc += (v[s][g].x + v[s][g].y) + (v[s][g].z + v[s][g].w);
        if (OP == 1) __builtin_nontemporal_store(-v[s][g], dst + (r + s) * rowvec + c + g * kSweepThreads);
        if (OP == 2) __builtin_nontemporal_store(v[s][g], dst + (r + s) * rowvec + c + g * kSweepThreads);
      }
  }
  if (OP == 0) {
    const double s = block_sum((double)acc, red);
    if (threadIdx.x == 0) sink[blockIdx.y * gridDim.x + blockIdx.x] = (float)s;
  }
}

template <int OP>
static int ceiling_launch(const void* srcv, void* dstv, size_t bytes, int64_t row_bytes, int map, float* sink, int blocks, hipStream_t st) {
  const nt_f4* src = static_cast<const nt_f4*>(srcv);
  nt_f4* dst = static_cast<nt_f4*>(dstv);
  const int64_t nvec = (int64_t)(bytes / 16);
  if (map == 0 || map == 1) {
    hipLaunchKernelGGL((ceiling_kernel<OP>), dim3(blocks), dim3(kSweepThreads), 0, st, src, dst, nvec, map == 0 ? 0 : row_bytes / 16, sink);
  } else if (map == 2) {
    const int64_t rowvec = row_bytes / 16, nrows = rowvec > 0 ? nvec / rowvec : 0;
    if (nrows <= 0 || rowvec % 1024 != 0) { set_error("ceiling: map 2 needs row_bytes = 16 KB * {1, 2, 4, 8}"); return CMTFPLS_EUNSUPPORTED; }
    switch (rowvec / 1024) {
      case 1: hipLaunchKernelGGL((ceiling_rowwg_kernel<OP, 1>), dim3(blocks), dim3(1024), 0, st, src, dst, nrows, sink); break;
      case 2: hipLaunchKernelGGL((ceiling_rowwg_kernel<OP, 2>), dim3(blocks), dim3(1024), 0, st, src, dst, nrows, sink); break;
      case 4: hipLaunchKernelGGL((ceiling_rowwg_kernel<OP, 4>), dim3(blocks), dim3(1024), 0, st, src, dst, nrows, sink); break;
      case 8: hipLaunchKernelGGL((ceiling_rowwg_kernel<OP, 8>), dim3(blocks), dim3(1024), 0, st, src, dst, nrows, sink); break;
      default: set_error("ceiling: map 2 needs row_bytes = 16 KB * {1, 2, 4, 8}"); return CMTFPLS_EUNSUPPORTED;
    }
  } else if (map == 3) {
    const int64_t rowvec = row_bytes / 16, nrows = rowvec > 0 ? nvec / rowvec : 0;
    if (nrows <= 0 || rowvec % (kSweepThreads * 2) != 0) { set_error("ceiling: map 3 needs row_bytes a multiple of 8 KB"); return CMTFPLS_EUNSUPPORTED; }
    const int ct = (int)(rowvec / (kSweepThreads * 2));
    int rb = blocks / ct;
    if (rb < 1) rb = 1;
    int64_t rpb = (nrows + rb - 1) / rb;
    rpb = (rpb + 3) / 4 * 4;                              // whole groups of 4 rows in flight
    rb = (int)((nrows + rpb - 1) / rpb);
    if ((int64_t)rb * rpb != nrows && nrows % 4 != 0) { set_error("ceiling: map 3 needs a row count that is a multiple of 4"); return CMTFPLS_EUNSUPPORTED; }
    hipLaunchKernelGGL((ceiling_colowner_kernel<OP>), dim3(ct, rb), dim3(kSweepThreads), 0, st, src, dst, nrows, rowvec, (int)rpb, sink);
  } else {
    set_error("ceiling: unknown map");
    return CMTFPLS_EINVAL;
  }
  return CMTFPLS_OK;
}

static int ceiling_args_ok(const void* a, const void* b, size_t bytes, int blocks, int64_t row_bytes) {
  return a && b && bytes >= 16 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 && blocks > 0 &&
         blocks <= 4096 && row_bytes >= 0 && (row_bytes % 16) == 0 && (row_bytes == 0 || (size_t)row_bytes <= bytes);
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

int cmtfpls_ceiling_max_blocks(void) { return 4096; }

int cmtfpls_ceiling_read(const void* buf, size_t bytes, int64_t row_bytes, int map, float* sink, int blocks, void* stream) {
  if (!ceiling_args_ok(buf, sink, bytes, blocks, row_bytes)) { set_error("ceiling_read: bad argument"); return CMTFPLS_EINVAL; }
  const int rc = ceiling_launch<0>(buf, nullptr, bytes, row_bytes, map, sink, blocks, (hipStream_t)stream);
  return rc != CMTFPLS_OK ? rc : check_launch("ceiling_read");
}

int cmtfpls_ceiling_rmw(void* buf, size_t bytes, int64_t row_bytes, int map, int blocks, void* stream) {
  if (!ceiling_args_ok(buf, buf, bytes, blocks, row_bytes)) { set_error("ceiling_rmw: bad argument"); return CMTFPLS_EINVAL; }
  const int rc = ceiling_launch<1>(buf, buf, bytes, row_bytes, map, nullptr, blocks, (hipStream_t)stream);
  return rc != CMTFPLS_OK ? rc : check_launch("ceiling_rmw");
}

int cmtfpls_ceiling_copy(const void* src, void* dst, size_t bytes, int64_t row_bytes, int map, int blocks, void* stream) {
  if (!ceiling_args_ok(src, dst, bytes, blocks, row_bytes) || src == dst) { set_error("ceiling_copy: bad argument"); return CMTFPLS_EINVAL; }
  const int rc = ceiling_launch<2>(src, dst, bytes, row_bytes, map, nullptr, blocks, (hipStream_t)stream);
  return rc != CMTFPLS_OK ? rc : check_launch("ceiling_copy");
}

}  // extern "C"
