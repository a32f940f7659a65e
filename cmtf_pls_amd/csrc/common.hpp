// Shared device helpers for the gfx950 NIPALS kernels (64-wide wavefronts throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cmtfpls.h"

namespace cmtfpls {

constexpr int kWave = 64;
// Every row-wise X sweep (center / score / deflate / score_deflate) is launched with exactly this many
// workgroups, grid-striding over rows.  512 x 256 threads = 2 workgroups = 8 rows in flight on each of
// the 256 CUs: measured best (6.9 TB/s against 6.6 TB/s with 2048; 256 is too few to cover the HBM
// latency) -- fewer concurrent row streams keep more DRAM pages open (profiles/r01p_tune_sweeps.txt).
#ifndef CMTFPLS_SWEEP_BLOCKS
#define CMTFPLS_SWEEP_BLOCKS 512
#endif
#ifndef CMTFPLS_UNROLL
#define CMTFPLS_UNROLL 4          // rows in flight per thread in the contraction
#endif
#ifndef CMTFPLS_ROW_UNROLL
#define CMTFPLS_ROW_UNROLL 8      // 16-byte loads in flight per lane in the wavefront-per-row sweeps (score, deflate)
#endif
#ifndef CMTFPLS_CONTRACT_BLOCKS
#define CMTFPLS_CONTRACT_BLOCKS 1024   // workgroups of the contraction (its partial rows = this / column tiles)
#endif
#ifndef CMTFPLS_NT_LOAD
#define CMTFPLS_NT_LOAD 1
#endif
#ifndef CMTFPLS_NT_STORE
#define CMTFPLS_NT_STORE 1
#endif
constexpr int kSweepBlocks = CMTFPLS_SWEEP_BLOCKS;
constexpr int kSweepThreads = 256;
constexpr int kUnroll = CMTFPLS_UNROLL;
constexpr int kRowUnroll = CMTFPLS_ROW_UNROLL;
constexpr int kContractBlocks = CMTFPLS_CONTRACT_BLOCKS;   // measured: profiles/r01m_tune_sweeps.txt

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

// Streaming accesses of X: every element is touched once per sweep, so loads and stores are marked
// non-temporal.  Measured on cfg-2 (profiles/r01c_tune_sweeps.txt): nt loads take the read sweeps from
// 5.76 / 5.86 TB/s to 6.42 / 6.53 TB/s (contraction / score); nt stores +2 % on the deflation sweep.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <int BYTES> struct NtRaw;
typedef double nt_d4 __attribute__((ext_vector_type(4)));
template <> struct NtRaw<32> { using type = nt_d4; };
template <> struct NtRaw<16> { using type = nt_f4; };
template <> struct NtRaw<8> { using type = double; };
template <> struct NtRaw<4> { using type = float; };

template <typename VT>
__device__ __forceinline__ VT ld_stream(const VT* p) {
#if CMTFPLS_NT_LOAD
  using R = typename NtRaw<sizeof(VT)>::type;
  const R r = __builtin_nontemporal_load(reinterpret_cast<const R*>(p));
  VT v;
  __builtin_memcpy(&v, &r, sizeof(VT));
  return v;
#else
  return *p;
#endif
}
template <typename VT>
__device__ __forceinline__ void st_stream(VT* p, const VT& v) {
#if CMTFPLS_NT_STORE
  using R = typename NtRaw<sizeof(VT)>::type;
  R r;
  __builtin_memcpy(&r, &v, sizeof(VT));
  __builtin_nontemporal_store(r, reinterpret_cast<R*>(p));
#else
  *p = v;
#endif
}

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

// Kronecker index walker: column c -> (j, k) = (c / B, c % B), advanced by a fixed stride
struct KronWalk {
  int j, k, dj, dk, B;
  __device__ __forceinline__ KronWalk(int64_t c0, int64_t stride, int B_) : B(B_) {
    j = (int)(c0 / B_);
    k = (int)(c0 % B_);
    dj = (int)(stride / B_);
    dk = (int)(stride % B_);
  }
  __device__ __forceinline__ void next() {
    k += dk;
    j += dj;
    if (k >= B) { k -= B; ++j; }
  }
};


// XCD-aware (column tile, row block) of a workgroup of a (col_tiles x row_blocks) grid.  Workgroups are dealt to the 8
// XCDs round-robin in dispatch order (linear id = blockIdx.x + gridDim.x * blockIdx.y), and every XCD has its own L2.
// The column tiles of ONE row block share per-row operands (the Y rows behind u = Y q, t, u): this map puts them on
// the SAME XCD (consecutive dispatch slots of that XCD), so those rows come from HBM once per row block instead of
// once per column tile.  Row blocks are taken in groups of 8 (one per XCD); a ragged tail keeps the plain order.
// A bijection for every grid shape.
struct TileId { int ct, rb; };
__device__ __forceinline__ TileId xcd_tile() {
  const unsigned CT = gridDim.x, RB = gridDim.y;
  const unsigned lin = blockIdx.x + CT * blockIdx.y;
  const unsigned G = CT * 8u, groups = RB / 8u;
  const unsigned grp = lin / G;
  TileId id;
  if (grp < groups) {
    const unsigned within = lin - grp * G;
    id.ct = (int)(within >> 3);
    id.rb = (int)(grp * 8u + (within & 7u));
  } else {
    const unsigned rest = lin - groups * G;
    id.ct = (int)(rest % CT);
    id.rb = (int)(groups * 8u + rest / CT);
  }
  return id;
}

// launch plan shared by the f64 and the mixed-precision cross-covariance kernels (xcov.hip, mixed.hip)
struct XcovPlan {
  int col_tiles, row_blocks, rows_per_block;
};
XcovPlan plan_xcov(int64_t I, int64_t P);
constexpr int kXcovMaxResponses = 64;       // accumulators of one pass (4 tiles of 16 per wavefront); more responses: more passes

void set_error(const char* msg);
int check_launch(const char* what);

}  // namespace cmtfpls
