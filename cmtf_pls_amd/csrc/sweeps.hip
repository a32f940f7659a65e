// X sweeps: the HBM-bound kernels of the NIPALS loop.  X is C-order (I, P), P = A*B, stored as
// f32 or f64; every accumulation is f64.  All reads/writes of X are 16-byte vectors along the
// contiguous trailing-mode fibre when the shape allows (B % (16/sizeof(T)) == 0), scalar otherwise.
//
//   colstats / mode0_contract : thread owns columns, loops over rows   -> (row-block x P) partials
//   center / score / deflate  : one wavefront per row, grid-stride     -> per-row wave reduction
//   score_deflate             : one workgroup per row, row kept in VGPRs between the two phases
//
// Loadings are passed factored (wA, wB); they are staged once per workgroup into LDS and the
// Khatri-Rao/Kronecker entry w[c] = wA[c / B] * wB[c % B] is formed on the fly.
#include "common.hpp"

namespace cmtfpls {

// ------------------------------------------------------------------------------------------
// mode-0 contraction / column statistics
// ------------------------------------------------------------------------------------------
#ifndef CMTFPLS_CONTRACT_U
#define CMTFPLS_CONTRACT_U 4   // 4 groups x 1024 workgroups: +2.8 % over 2 x 1024 at 65536 x 128 x 128 (profiles/r02d_tune_sweeps.txt)
#endif
constexpr int kContractU = CMTFPLS_CONTRACT_U;  // 16-byte column groups per thread
constexpr int kYqChunk = 2048;  // rows of u = Y q a workgroup keeps in LDS at a time (YQ variants)
#ifndef CMTFPLS_YQ_UNFUSE_TILES
#define CMTFPLS_YQ_UNFUSE_TILES 16
#endif
constexpr int kYqUnfuseTiles = CMTFPLS_YQ_UNFUSE_TILES;   // column tiles from which u = Y q is formed once, up front

struct ContractPlan {
  int vec;            // 1: vector kernel, 0: scalar kernel
  int col_tiles;      // gridDim.x
  int row_blocks;     // gridDim.y  (= number of partial rows in the workspace)
  int rows_per_block;
};

constexpr int kDcU = 2;   // column groups per thread of deflate_contract_kernel (it also keeps wB entries and writes back)

#ifndef CMTFPLS_CONTRACT_BLOCKS_FULL
#define CMTFPLS_CONTRACT_BLOCKS_FULL 512
#endif
#ifndef CMTFPLS_UNROLL_FULL
#define CMTFPLS_UNROLL_FULL 2
#endif
constexpr int kContractBlocksFull = CMTFPLS_CONTRACT_BLOCKS_FULL;   // workgroups of the guard-free (FULL) contraction: 7.0 TB/s against 6.0 with 1024
                                          // at 65536 x 128 x 128 (profiles/r02r_tune_full.txt); the guarded form keeps 1024

static ContractPlan plan_contract(int64_t I, int64_t P, int elem, int U = kContractU, int blocks = kContractBlocks) {
  ContractPlan p;
  const int V = 16 / elem;
  p.vec = (P % V == 0) ? 1 : 0;
  const int64_t tile = p.vec ? (int64_t)kSweepThreads * V * U : kSweepThreads;
  p.col_tiles = (int)((P + tile - 1) / tile);
  int64_t want = (blocks + p.col_tiles - 1) / p.col_tiles;            // ~`blocks` workgroups in all
  if (want < 1) want = 1;
  int64_t rpb = (I + want - 1) / want;
  if (rpb < 16) rpb = 16;                                           // amortise the partial store
  p.rows_per_block = (int)rpb;
  p.row_blocks = (int)((I + rpb - 1) / rpb);
  if (p.row_blocks < 1) p.row_blocks = 1;
  return p;
}

// u[i] = sum_m Y[i, m] q[m] for the rows [r0, r1) of one workgroup, into LDS (YQ variants: the Y
// update u = Y q of tpls.py:102 is formed where it is consumed instead of by a launch of its own).
// One thread per row, m ascending (the order of cmtfpls_rowdot_f64, bit for bit), 8 loads in flight.
// (A lane-group-per-row version with butterfly sums needed 40 more VGPRs for its shuffle indices and set
// the register count -- hence the occupancy -- of the whole contraction kernel.)
__device__ __forceinline__ void rows_times_q(const double* __restrict__ Y, int ldy, int M, const double* __restrict__ q,
                                             int64_t r0, int64_t r1, double* __restrict__ us) {
  const int nrows = (int)(r1 - r0);
  for (int rr = threadIdx.x; rr < nrows; rr += kSweepThreads) {
    const double* __restrict__ yr = Y + (r0 + rr) * ldy;
    double s = 0.0;
    int m = 0;
    for (; m + 8 <= M; m += 8) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = yr[m + k];
#pragma unroll
      for (int k = 0; k < 8; ++k) s = fma(v[k], q[m + k], s);
    }
    for (; m < M; ++m) s = fma(yr[m], q[m], s);
    us[rr] = s;
  }
  __syncthreads();
}

// MODE 0: plain (NaN propagates, as np.einsum)  1: NaN -> 0  2: statistics (u == 1, NaN -> 0, count)
// YQ: u is not read; u[i] = Y[i, :] . q is formed per workgroup in LDS, kYqChunk rows at a time
// FULL: every column group of every thread exists (P is a multiple of the column tile): no guards, so the 4 x U loads
// of a trip are issued back to back (with guards each load sits in its own exec-masked block).
// ILV (only with u read from memory): the row blocks are INTERLEAVED -- block rb takes rows rb, rb + RB, rb + 2 RB, ... --
// so that all workgroups advance through the tensor together, inside a window of RB x RU consecutive rows (16 MB at
// 256 x 256 f32) instead of RB fronts that lie I / RB rows (gigabytes) apart: on a 68.7 GB tensor the contiguous
// row blocks lose 5-10 % (profiles/r02bf_contract_forms_cfg5.txt: 6.2-6.4 -> 6.85-6.99 TB/s at 262144 x 256 x 256 inside
// bench.py, no difference at <= 131072 rows).  A read-only sibling of deflate_contract_rows_kernel (1024-thread workgroup
// per 64 KB row segment, rows interleaved by construction) was measured next to it: 6.35-6.43 TB/s there and 6.65-6.75
// at the smaller sizes, where this kernel reaches 7.0-7.2 -- not kept.
template <typename T, int MODE, bool YQ, int U, bool FULL, bool ILV = false>
__global__ __launch_bounds__(kSweepThreads) void contract_vec_kernel(
    const T* __restrict__ X, int64_t I, int64_t P, const double* __restrict__ u,
    double* __restrict__ part, double* __restrict__ cntpart, int rows_per_block,
    const double* __restrict__ Y, int ldy, int M, const double* __restrict__ q) {
  extern __shared__ double us[];
  constexpr int V = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  const TileId tile = xcd_tile();                       // the column tiles of a row block share one XCD's L2
  const int64_t cbase = (int64_t)tile.ct * (kSweepThreads * V * U) + (int64_t)threadIdx.x * V;
  static_assert(!(ILV && YQ), "interleaved row blocks read u from memory");
  const int64_t rs = ILV ? (int64_t)gridDim.y : 1;                           // row step (compile-time 1 without ILV)
  const int64_t r0 = ILV ? (int64_t)tile.rb : (int64_t)tile.rb * rows_per_block;
  const int64_t r1 = ILV ? I : ((r0 + rows_per_block < I) ? r0 + rows_per_block : I);
  double acc[U][V];
  double cnt[U][V];
  bool ok[U];
#pragma unroll
  for (int g = 0; g < U; ++g) {
    ok[g] = FULL || cbase + (int64_t)g * kSweepThreads * V < P;
#pragma unroll
    for (int e = 0; e < V; ++e) { acc[g][e] = 0.0; cnt[g][e] = 0.0; }
  }
  constexpr int RU = FULL ? CMTFPLS_UNROLL_FULL : kUnroll;   // rows in flight per thread (x U loads each)
  // YQ: the workgroup's rows go through LDS in chunks of kYqChunk rows of u (16 KB), whatever rows_per_block is
  const int64_t rend = r1;
  for (int64_t rc0 = r0; rc0 < rend; rc0 += (YQ ? (int64_t)kYqChunk : rend - r0)) {
  const int64_t r0 = rc0;                                                   // chunk = [r0, r1) below
  const int64_t r1 = (YQ && rc0 + kYqChunk < rend) ? rc0 + kYqChunk : rend;
  if (YQ) {
    if (rc0 != (int64_t)tile.rb * rows_per_block) __syncthreads();          // the previous chunk's readers are done
    rows_times_q(Y, ldy, M, q, r0, r1, us);
  }
  int64_t r = r0;
  for (; r + (RU - 1) * rs < r1; r += RU * rs) {
    VT x[RU][U];
    double uu[RU];
#pragma unroll
    for (int s = 0; s < RU; ++s) {
      uu[s] = (MODE == 2) ? 1.0 : YQ ? us[r + s - r0] : u[r + s * rs];
#pragma unroll
      for (int g = 0; g < U; ++g)
        if (FULL || ok[g]) x[s][g] = ld_stream(reinterpret_cast<const VT*>(X + (r + s * rs) * P + cbase + (int64_t)g * kSweepThreads * V));
    }
#pragma unroll
    for (int s = 0; s < RU; ++s)
#pragma unroll
      for (int g = 0; g < U; ++g)
        if (FULL || ok[g]) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T xv = x[s][g].e[e];
            if (MODE == 0) {
              acc[g][e] = fma((double)xv, uu[s], acc[g][e]);
            } else {
              const bool obs = (xv == xv);
              acc[g][e] = fma(obs ? (double)xv : 0.0, uu[s], acc[g][e]);
              if (MODE == 2) cnt[g][e] += obs ? 1.0 : 0.0;
            }
          }
        }
  }
  for (; r < r1; r += rs) {
    const double ur = (MODE == 2) ? 1.0 : YQ ? us[r - r0] : u[r];
#pragma unroll
    for (int g = 0; g < U; ++g)
      if (FULL || ok[g]) {
        const VT x = ld_stream(reinterpret_cast<const VT*>(X + r * P + cbase + (int64_t)g * kSweepThreads * V));
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T xv = x.e[e];
          if (MODE == 0) {
            acc[g][e] = fma((double)xv, ur, acc[g][e]);
          } else {
            const bool obs = (xv == xv);
            acc[g][e] = fma(obs ? (double)xv : 0.0, ur, acc[g][e]);
            if (MODE == 2) cnt[g][e] += obs ? 1.0 : 0.0;
          }
        }
      }
  }
  }   // chunks
#pragma unroll
  for (int g = 0; g < U; ++g)
    if (FULL || ok[g]) {
      const int64_t c = cbase + (int64_t)g * kSweepThreads * V;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        part[(int64_t)tile.rb * P + c + e] = acc[g][e];
        if (MODE == 2) cntpart[(int64_t)tile.rb * P + c + e] = cnt[g][e];
      }
    }
}

// Deflation of component a fused with the first contraction of component a+1 (tpls.py:109, then
// tpls.py:80-83 of the next pass of the component loop): X[i,c] -= t[i] wA[c/B] wB[c%B] is written back
// and the new value is contracted with the next u in the same sweep, so the first iteration of every
// component but the first costs one X read less.  Thread layout of contract_vec_kernel (a thread owns
// 2 x 16 B of columns: its wA / wB entries live in registers); the arithmetic per element is that of
// deflate_kernel followed by contract_vec_kernel, bit for bit; also the sum of squares of the new X.
template <typename T, int MODE, bool YQ>
__global__ __launch_bounds__(kSweepThreads) void deflate_contract_kernel(
    T* __restrict__ X, int64_t I, int64_t P, int B, const double* __restrict__ t,
    const double* __restrict__ wA, const double* __restrict__ wB, const double* __restrict__ u,
    double* __restrict__ part, double* __restrict__ ssq_part, int rows_per_block,
    const double* __restrict__ Y, int ldy, int M, const double* __restrict__ q) {
  extern __shared__ double us[];
  __shared__ double red[16];
  constexpr int V = VecOf<T>::N;
  constexpr int U = kDcU;
  using VT = typename VecOf<T>::type;
  const int64_t cbase = (int64_t)blockIdx.x * (kSweepThreads * V * U) + (int64_t)threadIdx.x * V;
  const int64_t rb0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t rend = (rb0 + rows_per_block < I) ? rb0 + rows_per_block : I;
  double acc[U][V], wb[U][V], wa[U];
  bool ok[U];
#pragma unroll
  for (int g = 0; g < U; ++g) {
    const int64_t c = cbase + (int64_t)g * kSweepThreads * V;
    ok[g] = c < P;
    const int64_t cs = ok[g] ? c : 0;
    wa[g] = wA[cs / B];                                  // B % V == 0: one j for the whole vector
#pragma unroll
    for (int e = 0; e < V; ++e) { acc[g][e] = 0.0; wb[g][e] = wB[cs % B + e]; }
  }
  double ssq = 0.0;
  constexpr int RU = kUnroll;
  for (int64_t rc0 = rb0; rc0 < rend; rc0 += (YQ ? (int64_t)kYqChunk : rend - rb0)) {
    const int64_t r0 = rc0;
    const int64_t r1 = (YQ && rc0 + kYqChunk < rend) ? rc0 + kYqChunk : rend;
    if (YQ) {
      if (rc0 != rb0) __syncthreads();
      rows_times_q(Y, ldy, M, q, r0, r1, us);
    }
    int64_t r = r0;
    for (; r + RU <= r1; r += RU) {
      VT x[RU][U];
      double uu[RU], tt[RU];
#pragma unroll
      for (int s = 0; s < RU; ++s) {
        uu[s] = YQ ? us[r + s - r0] : u[r + s];
        tt[s] = t[r + s];
#pragma unroll
        for (int g = 0; g < U; ++g)
          if (ok[g]) x[s][g] = ld_stream(reinterpret_cast<const VT*>(X + (r + s) * P + cbase + (int64_t)g * kSweepThreads * V));
      }
#pragma unroll
      for (int s = 0; s < RU; ++s)
#pragma unroll
        for (int g = 0; g < U; ++g)
          if (ok[g]) {
            const double tw = tt[s] * wa[g];
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const T nv = (T)fma(-tw, wb[g][e], (double)x[s][g].e[e]);
              x[s][g].e[e] = nv;
              const double d = (nv == nv) ? (double)nv : 0.0;          // NaN (missing) stays NaN, skipped in the norm
              ssq = fma(d, d, ssq);
              acc[g][e] = fma((MODE == 0) ? (double)nv : d, uu[s], acc[g][e]);
            }
            st_stream(reinterpret_cast<VT*>(X + (r + s) * P + cbase + (int64_t)g * kSweepThreads * V), x[s][g]);
          }
    }
    for (; r < r1; ++r) {
      const double ur = YQ ? us[r - r0] : u[r];
      const double tr = t[r];
#pragma unroll
      for (int g = 0; g < U; ++g)
        if (ok[g]) {
          VT x = ld_stream(reinterpret_cast<const VT*>(X + r * P + cbase + (int64_t)g * kSweepThreads * V));
          const double tw = tr * wa[g];
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T nv = (T)fma(-tw, wb[g][e], (double)x.e[e]);
            x.e[e] = nv;
            const double d = (nv == nv) ? (double)nv : 0.0;
            ssq = fma(d, d, ssq);
            acc[g][e] = fma((MODE == 0) ? (double)nv : d, ur, acc[g][e]);
          }
          st_stream(reinterpret_cast<VT*>(X + r * P + cbase + (int64_t)g * kSweepThreads * V), x);
        }
    }
  }
#pragma unroll
  for (int g = 0; g < U; ++g)
    if (ok[g]) {
      const int64_t c = cbase + (int64_t)g * kSweepThreads * V;
#pragma unroll
      for (int e = 0; e < V; ++e) part[(int64_t)blockIdx.y * P + c + e] = acc[g][e];
    }
  __syncthreads();                                       // `us` / `red` are not in use any more
  const double sblk = block_sum(ssq, red);
  if (threadIdx.x == 0) ssq_part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = sblk;
}

// The same fused sweep with one 1024-thread WORKGROUP per row segment (<= 1024 * V * 4 elements; one per CU): a
// lane owns the same 4 column vectors in every row its workgroup visits -- their wA / wB entries and the 16 f64
// accumulators of Z live in registers -- and two rows are in flight.  This is the access pattern that reaches the
// read-modify-write ceiling of the chip (6.0 TB/s, tools/exp/rowexp.hip "col-owner 1024thr x4 RU2"; the
// 256-thread column-tile form above stops at 5.0 TB/s under the register pressure of its 4 x 2 rows in flight).
// u = Y q is formed up front by the rowdot kernel (one small launch instead of a per-workgroup prologue).
// One partial row of Z per workgroup (gridDim / nseg partial rows), summed by reduce_rows_kernel in index order.
// FULL: every lane's 4 vectors exist (segment = 1024 * V * 4 elements exactly): no guards, straight-line code.
// MODE 0 (unmasked: the block has no missing value, Z propagates NaN like np.einsum) also drops the NaN test of
// the norm -- 3 of the 9 vector ALU operations per element.
template <typename T, int MODE, bool KC, bool FULL>
__global__ __launch_bounds__(1024) void deflate_contract_rows_kernel(
    T* __restrict__ X, int64_t I, unsigned P, int B, int nseg, const double* __restrict__ t,
    const double* __restrict__ wA, const double* __restrict__ wB, const double* __restrict__ u,
    double* __restrict__ part, double* __restrict__ ssq_part) {
  __shared__ double red[16];
  constexpr int V = VecOf<T>::N;
  constexpr int NV = 4, RU = 2;
  using VT = Pack<T, V>;
  constexpr unsigned stride = 1024u * V;
  const unsigned Pseg = P / (unsigned)nseg;
  const unsigned seg = blockIdx.x % (unsigned)nseg;
  const unsigned c0 = threadIdx.x * V;
  double acc[NV][V], wa[NV], wb[KC ? 1 : NV][V];
  bool ok[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const unsigned c = c0 + n * stride;
    ok[n] = FULL || c < Pseg;
    const unsigned cg = seg * Pseg + (ok[n] ? c : 0);
    wa[n] = wA[cg / (unsigned)B];                          // B % V == 0: one j for the whole vector
#pragma unroll
    for (int e = 0; e < V; ++e) {
      acc[n][e] = 0.0;
      if (!KC || n == 0) wb[KC ? 0 : n][e] = wB[cg % (unsigned)B + e];
    }
  }
  double ssq = 0.0;
  const int64_t step = gridDim.x / nseg;
  T* __restrict__ xs = X + seg * Pseg + c0;
  int64_t r = blockIdx.x / nseg;
  for (; r + (RU - 1) * step < I; r += RU * step) {
    VT x[RU][NV];
    double uu[RU], tt[RU];
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      uu[q] = u[r + q * step];
      tt[q] = t[r + q * step];
#pragma unroll
      for (int n = 0; n < NV; ++n)
        if (FULL || ok[n]) x[q][n] = ld_stream(reinterpret_cast<const VT*>(xs + (r + q * step) * (int64_t)P + n * stride));
    }
#pragma unroll
    for (int q = 0; q < RU; ++q)
#pragma unroll
      for (int n = 0; n < NV; ++n)
        if (FULL || ok[n]) {
          const double tw = tt[q] * wa[n];
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T nv = (T)fma(-tw, wb[KC ? 0 : n][e], (double)x[q][n].e[e]);
            x[q][n].e[e] = nv;
            const double d = (MODE == 0 || nv == nv) ? (double)nv : 0.0;   // NaN (missing) stays NaN, skipped in the norm
            ssq = fma(d, d, ssq);
            acc[n][e] = fma((MODE == 0) ? (double)nv : d, uu[q], acc[n][e]);
          }
          st_stream(reinterpret_cast<VT*>(xs + (r + q * step) * (int64_t)P + n * stride), x[q][n]);
        }
  }
  for (; r < I; r += step) {
    const double ur = u[r], tr = t[r];
#pragma unroll
    for (int n = 0; n < NV; ++n)
      if (FULL || ok[n]) {
        VT x = ld_stream(reinterpret_cast<const VT*>(xs + r * (int64_t)P + n * stride));
        const double tw = tr * wa[n];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T nv = (T)fma(-tw, wb[KC ? 0 : n][e], (double)x.e[e]);
          x.e[e] = nv;
          const double d = (MODE == 0 || nv == nv) ? (double)nv : 0.0;
          ssq = fma(d, d, ssq);
          acc[n][e] = fma((MODE == 0) ? (double)nv : d, ur, acc[n][e]);
        }
        st_stream(reinterpret_cast<VT*>(xs + r * (int64_t)P + n * stride), x);
      }
  }
  double* __restrict__ prow = part + (int64_t)(blockIdx.x / nseg) * P + seg * Pseg + c0;
#pragma unroll
  for (int n = 0; n < NV; ++n)
    if (FULL || ok[n]) {
#pragma unroll
      for (int e = 0; e < V; ++e) prow[n * stride + e] = acc[n][e];
    }
  const double sblk = block_sum(ssq, red);
  if (threadIdx.x == 0) ssq_part[blockIdx.x] = sblk;
}

constexpr int kDcRowsGrid = 256;        // one 1024-thread workgroup per CU

// segments per row for the workgroup-per-row-segment kernels (0: shape outside that form)
static int rows_nseg(int64_t P, int V, int grid) {
  if (P <= (int64_t)256 * V * 4 || P >= ((int64_t)1 << 31)) return 0;
  const int64_t segmax = (int64_t)1024 * V * 4;
  int nseg = (int)((P + segmax - 1) / segmax);
  while (nseg <= 64 && ((P % ((int64_t)nseg * V)) != 0 || (grid % nseg) != 0)) ++nseg;
  return nseg <= 64 ? nseg : 0;
}

// Narrow blocks (P <= 128 vectors of 16 B, e.g. the I x 512 matrix block of a coupled fit): one row
// needs only ncv = P / V threads, so the workgroup's 256 threads take RS = 256 / ncv rows at a time
// (thread = (row lane, column vector)) and the RS row lanes are added in index order through LDS.
// Same partial layout and same second kernel as contract_vec_kernel.
template <typename T, int MODE, bool YQ>
__global__ __launch_bounds__(kSweepThreads) void contract_narrow_kernel(
    const T* __restrict__ X, int64_t I, int64_t P, const double* __restrict__ u,
    double* __restrict__ part, double* __restrict__ cntpart, int rows_per_block,
    const double* __restrict__ Y, int ldy, int M, const double* __restrict__ q, int ncv, int RS) {
  extern __shared__ double us[];                       // YQ: kYqChunk doubles
  constexpr int V = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  __shared__ double red[kSweepThreads * V];             // [row lane][column] = RS x P <= 256 V doubles
  __shared__ double redc[(MODE == 2) ? kSweepThreads * V : 1];
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < I) ? r0 + rows_per_block : I;
  const int cv = threadIdx.x % ncv, rl = threadIdx.x / ncv;
  const bool active = rl < RS;
  const int64_t c = (int64_t)cv * V;
  double acc[V], cnt[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { acc[e] = 0.0; cnt[e] = 0.0; }
  const int64_t rend = r1;
  for (int64_t rc0 = r0; rc0 < rend; rc0 += (YQ ? (int64_t)kYqChunk : rend - r0)) {   // chunks of u in LDS (YQ)
  const int64_t r0 = rc0;
  const int64_t r1 = (YQ && rc0 + kYqChunk < rend) ? rc0 + kYqChunk : rend;
  if (YQ) {
    if (rc0 != (int64_t)blockIdx.y * rows_per_block) __syncthreads();
    rows_times_q(Y, ldy, M, q, r0, r1, us);
  }
  if (active) {
    constexpr int RU = kUnroll;
    int64_t r = r0 + rl;
    for (; r + (int64_t)(RU - 1) * RS < r1; r += (int64_t)RU * RS) {
      VT x[RU];
      double uu[RU];
#pragma unroll
      for (int s = 0; s < RU; ++s) {
        const int64_t rr = r + (int64_t)s * RS;
        uu[s] = (MODE == 2) ? 1.0 : YQ ? us[rr - r0] : u[rr];
        x[s] = ld_stream(reinterpret_cast<const VT*>(X + rr * P + c));
      }
#pragma unroll
      for (int s = 0; s < RU; ++s)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T xv = x[s].e[e];
          if (MODE == 0) {
            acc[e] = fma((double)xv, uu[s], acc[e]);
          } else {
            const bool obs = (xv == xv);
            acc[e] = fma(obs ? (double)xv : 0.0, uu[s], acc[e]);
            if (MODE == 2) cnt[e] += obs ? 1.0 : 0.0;
          }
        }
    }
    for (; r < r1; r += RS) {
      const double ur = (MODE == 2) ? 1.0 : YQ ? us[r - r0] : u[r];
      const VT x = ld_stream(reinterpret_cast<const VT*>(X + r * P + c));
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T xv = x.e[e];
        if (MODE == 0) {
          acc[e] = fma((double)xv, ur, acc[e]);
        } else {
          const bool obs = (xv == xv);
          acc[e] = fma(obs ? (double)xv : 0.0, ur, acc[e]);
          if (MODE == 2) cnt[e] += obs ? 1.0 : 0.0;
        }
      }
    }
  }
  }   // chunks
  if (active) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red[(int64_t)rl * P + c + e] = acc[e];
      if (MODE == 2) redc[(int64_t)rl * P + c + e] = cnt[e];
    }
  }
  __syncthreads();
  for (int64_t cc = threadIdx.x; cc < P; cc += kSweepThreads) {
    double tot = 0.0, totc = 0.0;
    for (int g = 0; g < RS; ++g) {
      tot += red[(int64_t)g * P + cc];
      if (MODE == 2) totc += redc[(int64_t)g * P + cc];
    }
    part[(int64_t)blockIdx.y * P + cc] = tot;
    if (MODE == 2) cntpart[(int64_t)blockIdx.y * P + cc] = totc;
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(kSweepThreads) void contract_scalar_kernel(
    const T* __restrict__ X, int64_t I, int64_t P, const double* __restrict__ u,
    double* __restrict__ part, double* __restrict__ cntpart, int rows_per_block) {
  const int64_t c = (int64_t)blockIdx.x * kSweepThreads + threadIdx.x;
  if (c >= P) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < I) ? r0 + rows_per_block : I;
  double acc = 0.0, cnt = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const T xv = X[r * P + c];
    const double us = (MODE == 2) ? 1.0 : u[r];
    if (MODE == 0) {
      acc = fma((double)xv, us, acc);
    } else {
      const bool obs = (xv == xv);
      acc = fma(obs ? (double)xv : 0.0, us, acc);
      if (MODE == 2) cnt += obs ? 1.0 : 0.0;
    }
  }
  part[(int64_t)blockIdx.y * P + c] = acc;
  if (MODE == 2) cntpart[(int64_t)blockIdx.y * P + c] = cnt;
}

// out[c] = sum_r part[r][c]: 32 columns x 8 row groups per workgroup, each thread sums every 8th
// partial row of its column, the 8 group sums are then added in a fixed order (bit-reproducible).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double* __restrict__ part, int nrows,
                                                         int64_t P, double* __restrict__ out) {
  __shared__ double red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 32 + cx;
  double s = 0.0;
  if (c < P) {
    // batches of 8 independent loads (clamped row, masked value), added in the same ascending order
    const double* __restrict__ pc = part + c;
    for (int r = ry; r < nrows; r += 64) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rr = r + 8 * k;
        v[k] = pc[(int64_t)((rr < nrows) ? rr : nrows - 1) * P];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += (r + 8 * k < nrows) ? v[k] : 0.0;
    }
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < P) {
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) tot += red[g][cx];
    out[c] = tot;
  }
}

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st) {
  const int grid = (int)((P + 31) / 32);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(grid), dim3(256), 0, st, part, nrows, P, out);
}

// Y != nullptr: the YQ form (u = Y q formed in the kernel; u itself is not read)
template <typename T, int MODE>
static int run_contract(const T* X, int64_t I, int64_t P, const double* u, double* out, double* cnt_out,
                        void* ws, size_t ws_bytes, hipStream_t st,
                        const double* Y = nullptr, int ldy = 0, int M = 0, const double* q = nullptr) {
  const bool yq = Y != nullptr;
  if (!X || !out || I <= 0 || P <= 0 || (MODE != 2 && !u && !yq) || (MODE == 2 && !cnt_out) || (yq && (!q || M <= 0 || ldy < M))) {
    set_error("mode0_contract/colstats: bad argument");
    return CMTFPLS_EINVAL;
  }
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) { set_error("X must be 16-byte aligned"); return CMTFPLS_EINVAL; }
  // column groups per thread: 4 (wider tiles, half as many partial rows per column: +2.8 % at 65536 rows) when a
  // workgroup still gets >= 128 rows, else 2 (shards of <= 16 K rows: 94 vs 101 us at 8192 x 128 x 128,
  // profiles/r02g_tune_small.txt); the statistics pass carries a second set of accumulators and stays at 2
  constexpr int Vp = 16 / (int)sizeof(T);
  const bool vecp = (P % Vp) == 0;
  const bool full4 = vecp && (P % ((int64_t)kSweepThreads * Vp * 4)) == 0, full2 = vecp && (P % ((int64_t)kSweepThreads * Vp * 2)) == 0;
  // very long rows (>= 16 column tiles of 4 groups, e.g. 256 x 256 f32): one workgroup per CU is best (7.2 against 6.95 TB/s
  // at 32768 x 256 x 256, profiles/r02t_tune_full.txt)
  const int64_t tiles4 = vecp ? (P + (int64_t)kSweepThreads * Vp * 4 - 1) / ((int64_t)kSweepThreads * Vp * 4) : 0;
  const bool one_per_cu = tiles4 >= 16 && I * tiles4 <= (int64_t)8192 * (kContractBlocksFull / 2);   // ... while a workgroup's
  // row block stays <= 8192 rows: at 262144 x 256 x 256, 256 workgroups fall to 6.8 TB/s and 512 hold 7.06 (profiles/r02ad_tune_cfg5.txt)
  const int blocks4 = !full4 ? kContractBlocks : (one_per_cu ? kContractBlocksFull / 2 : kContractBlocksFull);
  const ContractPlan p4 = plan_contract(I, P, (int)sizeof(T), 4, blocks4);
  const bool wideU = MODE != 2 && kContractU == 4 && p4.rows_per_block >= 128;
  const ContractPlan p = wideU ? p4 : plan_contract(I, P, (int)sizeof(T), 2, full2 ? kContractBlocksFull : kContractBlocks);
  const bool fullt = wideU ? full4 : full2;
#define CV_LAUNCH1(YQF, UU, FF, LDS, UPTR, YP, LDY, MM, QP)                                                                          \
  do {                                                                                                                               \
    if (!YQF && ilv)                                                                                                                 \
      hipLaunchKernelGGL((contract_vec_kernel<T, MODE, false, UU, FF, true>), grid, dim3(kSweepThreads), LDS, st, X, I, P, UPTR,      \
                         part, cntpart, p.rows_per_block, YP, LDY, MM, QP);                                                          \
    else                                                                                                                             \
      hipLaunchKernelGGL((contract_vec_kernel<T, MODE, YQF, UU, FF, false>), grid, dim3(kSweepThreads), LDS, st, X, I, P, UPTR,       \
                         part, cntpart, p.rows_per_block, YP, LDY, MM, QP);                                                          \
  } while (0)
#define CV_LAUNCH(YQF, LDS, UPTR, YP, LDY, MM, QP)                                                                                   \
  do {                                                                                                                               \
    if (wideU && fullt) CV_LAUNCH1(YQF, ((MODE == 2) ? 2 : 4), true, LDS, UPTR, YP, LDY, MM, QP);                                     \
    else if (wideU) CV_LAUNCH1(YQF, ((MODE == 2) ? 2 : 4), false, LDS, UPTR, YP, LDY, MM, QP);                                        \
    else if (fullt) CV_LAUNCH1(YQF, 2, true, LDS, UPTR, YP, LDY, MM, QP);                                                            \
    else CV_LAUNCH1(YQF, 2, false, LDS, UPTR, YP, LDY, MM, QP);                                                                      \
  } while (0)
  // Wide blocks (>= kYqUnfuseTiles column tiles): every column tile of a row block would repeat the same
  // u = Y q prologue, so u is formed once by the rowdot kernel into the tail of the workspace instead
  // (262144 x 256 x 256, M = 32: 2 % of the sweep)
  const bool yq_pre = yq && p.vec && M <= 64 && MODE != 2 && p.col_tiles >= kYqUnfuseTiles;
  // interleaved row blocks (see contract_vec_kernel) for wide rows whose u comes from memory
#ifndef CMTFPLS_CONTRACT_ILV
#define CMTFPLS_CONTRACT_ILV 1
#endif
  const bool ilv = MODE != 2 && p.vec && (CMTFPLS_CONTRACT_ILV == 2 || (CMTFPLS_CONTRACT_ILV == 1 && p.col_tiles >= kYqUnfuseTiles));
  const size_t need = (size_t)p.row_blocks * (size_t)P * sizeof(double) * (MODE == 2 ? 2 : 1) + (yq_pre ? (size_t)I * sizeof(double) : 0);
  if (!ws || ws_bytes < need) { set_error("mode0_contract/colstats: workspace too small"); return CMTFPLS_EWORKSPACE; }
  double* part = static_cast<double*>(ws);
  double* cntpart = (MODE == 2) ? part + (size_t)p.row_blocks * P : nullptr;
  const dim3 grid(p.col_tiles, p.row_blocks);
  constexpr int Vt = 16 / (int)sizeof(T);
  const int ncv = p.vec ? (int)(P / Vt) : 0;
  const bool narrow = p.vec && ncv <= kSweepThreads / 2;          // at least two rows per workgroup pass
  const int RS = narrow ? kSweepThreads / ncv : 1;
  if (yq_pre) {
    double* u_ws = part + (size_t)p.row_blocks * P;
    const int rc = cmtfpls_rowdot_f64(Y, ldy, M, I, q, u_ws, nullptr, nullptr, nullptr, 0, st);
    if (rc != CMTFPLS_OK) return rc;
    CV_LAUNCH(false, 0, u_ws, nullptr, 0, 0, nullptr);
  } else if (yq) {
    // supported: vector shape, M <= 64 (one Y row per wavefront pass)
    const size_t lds = (size_t)kYqChunk * sizeof(double);
    if (!p.vec || M > 64 || MODE == 2) {
      set_error("mode0_contract_yq: shape outside the fused form; form u = Y q with rowdot and use mode0_contract");
      return CMTFPLS_EUNSUPPORTED;
    }
    if (narrow)
      hipLaunchKernelGGL((contract_narrow_kernel<T, MODE, true>), grid, dim3(kSweepThreads), lds, st, X, I, P, u, part, cntpart,
                         p.rows_per_block, Y, ldy, M, q, ncv, RS);
    else
      CV_LAUNCH(true, lds, u, Y, ldy, M, q);
  } else if (narrow) {
    hipLaunchKernelGGL((contract_narrow_kernel<T, MODE, false>), grid, dim3(kSweepThreads), 0, st, X, I, P, u, part, cntpart,
                       p.rows_per_block, nullptr, 0, 0, nullptr, ncv, RS);
  } else if (p.vec) {
    CV_LAUNCH(false, 0, u, nullptr, 0, 0, nullptr);
  } else {
    hipLaunchKernelGGL((contract_scalar_kernel<T, MODE>), grid, dim3(kSweepThreads), 0, st, X, I, P, u, part, cntpart, p.rows_per_block);
  }
  launch_reduce_rows(part, p.row_blocks, P, out, st);
  if (MODE == 2) launch_reduce_rows(cntpart, p.row_blocks, P, cnt_out, st);
#undef CV_LAUNCH
#undef CV_LAUNCH1
  return check_launch("mode0_contract");
}

template <typename T>
static int run_deflate_contract(T* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB,
                                const double* Y, int ldy, int M, const double* q, double* Z, int masked, double* ssq,
                                void* ws, size_t ws_bytes, hipStream_t st) {
  if (!X || !t || !wA || !wB || !Y || !q || !Z || !ssq || I <= 0 || A <= 0 || B <= 0 || M <= 0 || ldy < M) {
    set_error("deflate_contract_yq: bad argument");
    return CMTFPLS_EINVAL;
  }
  const int64_t P = (int64_t)A * B;
  constexpr int Vt = 16 / (int)sizeof(T);
  const ContractPlan p = plan_contract(I, P, (int)sizeof(T), kDcU);
  if (!p.vec || (B % Vt) != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 || M > 64) {
    set_error("deflate_contract_yq: shape outside the fused form; use deflate, then mode0_contract");
    return CMTFPLS_EUNSUPPORTED;
  }
#ifndef CMTFPLS_DC_ROWS
#define CMTFPLS_DC_ROWS 1
#endif
  const int nseg = CMTFPLS_DC_ROWS ? rows_nseg(P, Vt, kDcRowsGrid) : 0;
  if (nseg > 0 && I >= 2 * kDcRowsGrid) {
    // workgroup-per-row-segment form: partial rows | ssq partials | u = Y q
    const size_t nrows_part = (size_t)(kDcRowsGrid / nseg);
    const size_t need_r = (nrows_part * (size_t)P + (size_t)kDcRowsGrid + (size_t)I) * sizeof(double);
    if (!ws || ws_bytes < need_r) { set_error("deflate_contract_yq: workspace too small"); return CMTFPLS_EWORKSPACE; }
    double* part = static_cast<double*>(ws);
    double* sspart = part + nrows_part * (size_t)P;
    double* u_ws = sspart + kDcRowsGrid;
    const int rc = cmtfpls_rowdot_f64(Y, ldy, M, I, q, u_ws, nullptr, nullptr, nullptr, 0, st);
    if (rc != CMTFPLS_OK) return rc;
    const bool kc = ((1024 * Vt) % B) == 0 && ((P / nseg) % B) == 0;
    const dim3 g(kDcRowsGrid), b(1024);
    const bool full = kc && (P / nseg) == (int64_t)1024 * Vt * 4;
#define DCR(MD, K, F) hipLaunchKernelGGL((deflate_contract_rows_kernel<T, MD, K, F>), g, b, 0, st, X, I, (unsigned)P, B, nseg, t, wA, wB, u_ws, part, sspart)
    if (masked) { if (full) DCR(1, true, true); else if (kc) DCR(1, true, false); else DCR(1, false, false); }
    else        { if (full) DCR(0, true, true); else if (kc) DCR(0, true, false); else DCR(0, false, false); }
#undef DCR
    launch_reduce_rows(part, (int)nrows_part, P, Z, st);
    launch_reduce_rows(sspart, kDcRowsGrid, 1, ssq, st);
    return check_launch("deflate_contract_yq");
  }
  const size_t nss = (size_t)p.col_tiles * p.row_blocks;
  const bool pre = p.col_tiles >= kYqUnfuseTiles;          // wide block: u = Y q once, up front (see run_contract)
  const size_t need = ((size_t)p.row_blocks * (size_t)P + nss + (pre ? (size_t)I : 0)) * sizeof(double);
  if (!ws || ws_bytes < need) { set_error("deflate_contract_yq: workspace too small"); return CMTFPLS_EWORKSPACE; }
  double* part = static_cast<double*>(ws);
  double* sspart = part + (size_t)p.row_blocks * P;
  const dim3 grid(p.col_tiles, p.row_blocks);
  const size_t lds = (size_t)kYqChunk * sizeof(double);
  if (pre) {
    double* u_ws = sspart + nss;
    const int rc = cmtfpls_rowdot_f64(Y, ldy, M, I, q, u_ws, nullptr, nullptr, nullptr, 0, st);
    if (rc != CMTFPLS_OK) return rc;
    if (masked)
      hipLaunchKernelGGL((deflate_contract_kernel<T, 1, false>), grid, dim3(kSweepThreads), 0, st, X, I, P, B, t, wA, wB, u_ws,
                         part, sspart, p.rows_per_block, nullptr, 0, 0, nullptr);
    else
      hipLaunchKernelGGL((deflate_contract_kernel<T, 0, false>), grid, dim3(kSweepThreads), 0, st, X, I, P, B, t, wA, wB, u_ws,
                         part, sspart, p.rows_per_block, nullptr, 0, 0, nullptr);
  } else if (masked)
    hipLaunchKernelGGL((deflate_contract_kernel<T, 1, true>), grid, dim3(kSweepThreads), lds, st, X, I, P, B, t, wA, wB, nullptr,
                       part, sspart, p.rows_per_block, Y, ldy, M, q);
  else
    hipLaunchKernelGGL((deflate_contract_kernel<T, 0, true>), grid, dim3(kSweepThreads), lds, st, X, I, P, B, t, wA, wB, nullptr,
                       part, sspart, p.rows_per_block, Y, ldy, M, q);
  launch_reduce_rows(part, p.row_blocks, P, Z, st);
  launch_reduce_rows(sspart, (int)nss, 1, ssq, st);
  return check_launch("deflate_contract_yq");
}

__device__ __forceinline__ void stage_loadings(double* sA, double* sB, const double* __restrict__ wA,
                                               const double* __restrict__ wB, int A, int B) {
  for (int i = threadIdx.x; i < A; i += blockDim.x) sA[i] = wA[i];
  for (int i = threadIdx.x; i < B; i += blockDim.x) sB[i] = wB[i];
  __syncthreads();
}
static inline size_t loadings_lds_bytes(int A, int B) { return ((size_t)((A + 1) & ~1) + (size_t)((B + 1) & ~1)) * sizeof(double); }

// sum_e x[e] * wB[k + e]   (V consecutive k: never straddles a j boundary because B % V == 0)
template <typename T, int V, bool MASKED>
__device__ __forceinline__ double dot_pack(const Pack<T, V>& x, const double* sBk) {
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    T xv = x.e[e];
    if (MASKED) xv = (xv == xv) ? xv : (T)0;
    s = fma((double)xv, sBk[e], s);
  }
  return s;
}

// ------------------------------------------------------------------------------------------
// score: t[i] = sum_c X[i,c] wA[c/B] wB[c%B]      one wavefront per row
// ------------------------------------------------------------------------------------------
// GRAM: also the partial sums of Y^T t (tpls.py:100) of this workgroup's rows: lane m < M of every
// wavefront accumulates t[i] * Y[i, m] over its rows, the wavefronts are added in index order and the
// workgroup writes qpart[blockIdx.x, 0:M]; a small kernel adds the workgroups in index order.
// GL: the loadings do not fit the LDS (A + B > 12 K doubles, e.g. a matrix block with > 12 K columns, A = 1):
// they are read from global memory (L2-resident: at most a few hundred KB shared by every workgroup).
template <typename T, bool MASKED, bool VEC, bool GRAM, bool GL = false>
__global__ __launch_bounds__(kSweepThreads) __attribute__((amdgpu_waves_per_eu(5))) void score_kernel(
    const T* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ wA,
    const double* __restrict__ wB, const double* __restrict__ rowcnt, double* __restrict__ t,
    const double* __restrict__ Y, int ldy, int M, double* __restrict__ qpart) {
  extern __shared__ double lds[];
  __shared__ double qs[kSweepThreads / kWave][kWave];
  double qacc = 0.0;
  const double* sA = GL ? wA : lds;
  const double* sB = GL ? wB : lds + ((A + 1) & ~1);
  if (!GL) stage_loadings(lds, lds + ((A + 1) & ~1), wA, wB, A, B);
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  const int lane = threadIdx.x & 63;
  const int64_t P = (int64_t)A * B;
  const int64_t nwaves = (int64_t)gridDim.x * (kSweepThreads / kWave);
  const int64_t step = (int64_t)kWave * V;
  const int64_t c0 = (int64_t)lane * V;
  const KronWalk w0(c0, step, B);
  for (int64_t row = (int64_t)blockIdx.x * (kSweepThreads / kWave) + (threadIdx.x >> 6); row < I; row += nwaves) {
    const T* __restrict__ xr = X + row * P;
    double yv = 0.0;
    if (GRAM) yv = Y[row * ldy + ((lane < M) ? lane : 0)];     // in flight while the row streams
    double acc = 0.0;
    KronWalk w = w0;
    int64_t c = c0;
    constexpr int UN = kRowUnroll;
    for (; c + (UN - 1) * step < P; c += UN * step) {
      VT x[UN];
#pragma unroll
      for (int s = 0; s < UN; ++s) x[s] = ld_stream(reinterpret_cast<const VT*>(xr + c + s * step));
#pragma unroll
      for (int s = 0; s < UN; ++s) {
        acc = fma(sA[w.j], dot_pack<T, V, MASKED>(x[s], sB + w.k), acc);
        w.next();
      }
    }
    for (; c < P; c += step) {
      const VT x = ld_stream(reinterpret_cast<const VT*>(xr + c));
      acc = fma(sA[w.j], dot_pack<T, V, MASKED>(x, sB + w.k), acc);
      w.next();
    }
    acc = wave_sum(acc);                                        // identical bits in every lane
    const double ti = MASKED ? acc / rowcnt[row] * (double)P : acc;
    if (lane == 0) t[row] = ti;
    if (GRAM) qacc = fma(ti, yv, qacc);
  }
  if (GRAM) {
    qs[threadIdx.x >> 6][lane] = qacc;
    __syncthreads();
    if (threadIdx.x < M) {
      double tot = 0.0;
#pragma unroll
      for (int wv = 0; wv < kSweepThreads / kWave; ++wv) tot += qs[wv][threadIdx.x];
      qpart[(int64_t)blockIdx.x * M + threadIdx.x] = tot;
    }
  }
}

// score of a FEW LONG rows (I <= 64 rows of >= 8192 elements: S = Y^T X_(0) with its M rows inside the cross-covariance
// loop, a handful of new samples in transform): the wavefront-per-row kernel above keeps I wavefronts busy -- 16 for the
// 16 x 16384 S of BASELINE configs[1], 13 us per call -- so here a 1024-thread workgroup takes one row: every thread its vectors
// at stride 1024 V, per-lane sequential sum, butterfly wave sum, the 16 wavefronts added in index order.
template <typename T, bool MASKED>
__global__ __launch_bounds__(1024) void score_fewrows_kernel(const T* __restrict__ X, int A, int B, const double* __restrict__ wA,
                                                            const double* __restrict__ wB, const double* __restrict__ rowcnt,
                                                            double* __restrict__ t) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  const double* sA = lds;
  const double* sB = lds + ((A + 1) & ~1);
  stage_loadings(lds, lds + ((A + 1) & ~1), wA, wB, A, B);
  constexpr int V = VecOf<T>::N;
  using VT = Pack<T, V>;
  const int64_t P = (int64_t)A * B;
  const int64_t step = (int64_t)1024 * V;
  const int64_t row = blockIdx.x;
  const T* __restrict__ xr = X + row * P;
  KronWalk w((int64_t)threadIdx.x * V, step, B);
  double acc = 0.0;
  int64_t c = (int64_t)threadIdx.x * V;
  constexpr int UN = 4;
  for (; c + (UN - 1) * step < P; c += UN * step) {
    VT x[UN];
#pragma unroll
    for (int s = 0; s < UN; ++s) x[s] = ld_stream(reinterpret_cast<const VT*>(xr + c + s * step));
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      acc = fma(sA[w.j], dot_pack<T, V, MASKED>(x[s], sB + w.k), acc);
      w.next();
    }
  }
  for (; c < P; c += step) {
    const VT x = ld_stream(reinterpret_cast<const VT*>(xr + c));
    acc = fma(sA[w.j], dot_pack<T, V, MASKED>(x, sB + w.k), acc);
    w.next();
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) t[row] = MASKED ? acc / rowcnt[row] * (double)P : acc;
}

// ------------------------------------------------------------------------------------------
// deflate: X[i,c] -= t[i] wA[c/B] wB[c%B]  (+ sum of squares of what is left)   wavefront per row
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC, bool GL = false>
__global__ __launch_bounds__(kSweepThreads) void deflate_kernel(
    T* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ t,
    const double* __restrict__ wA, const double* __restrict__ wB, double* __restrict__ ssq_part) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  const double* sA = GL ? wA : lds;                    // GL: loadings too long for the LDS, read through L2
  const double* sB = GL ? wB : lds + ((A + 1) & ~1);
  if (!GL) stage_loadings(lds, lds + ((A + 1) & ~1), wA, wB, A, B);
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  const int lane = threadIdx.x & 63;
  const int64_t P = (int64_t)A * B;
  const int64_t nwaves = (int64_t)gridDim.x * (kSweepThreads / kWave);
  const int64_t step = (int64_t)kWave * V;
  const int64_t c0 = (int64_t)lane * V;
  const KronWalk w0(c0, step, B);
  double ssq = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * (kSweepThreads / kWave) + (threadIdx.x >> 6); row < I; row += nwaves) {
    T* __restrict__ xr = X + row * P;
    const double ti = t[row];
    KronWalk w = w0;
    int64_t c = c0;
    constexpr int UN = kRowUnroll;
    for (; c + (UN - 1) * step < P; c += UN * step) {
      VT x[UN];
#pragma unroll
      for (int s = 0; s < UN; ++s) x[s] = ld_stream(reinterpret_cast<const VT*>(xr + c + s * step));
#pragma unroll
      for (int s = 0; s < UN; ++s) {
        const double tw = ti * sA[w.j];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T nv = (T)fma(-tw, sB[w.k + e], (double)x[s].e[e]);
          x[s].e[e] = nv;
          const double d = (nv == nv) ? (double)nv : 0.0;   // NaN (missing) stays NaN, skipped in the norm
          ssq = fma(d, d, ssq);
        }
        st_stream(reinterpret_cast<VT*>(xr + c + s * step), x[s]);
        w.next();
      }
    }
    for (; c < P; c += step) {
      VT x = ld_stream(reinterpret_cast<const VT*>(xr + c));
      const double tw = ti * sA[w.j];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T nv = (T)fma(-tw, sB[w.k + e], (double)x.e[e]);
        x.e[e] = nv;
        const double d = (nv == nv) ? (double)nv : 0.0;
        ssq = fma(d, d, ssq);
      }
      st_stream(reinterpret_cast<VT*>(xr + c), x);
      w.next();
    }
  }
  if (ssq_part) {
    const double s = block_sum(ssq, red);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = s;
  }
}

// Deflation with one WORKGROUP per row (vector shapes, rows of up to MAXT * V * NV elements): the whole
// row is read (NV 16-byte loads per lane in flight), updated and written back by the same workgroup, so
// HBM sees long read bursts followed by long write bursts of the same pages.  Measured against the
// wavefront-per-row kernel above: 1.57 vs 1.60 ms at 65536 x 128 x 128 f32, 3.15 vs 3.27 ms at
// 32768 x 256 x 256 (profiles/r01p_tune_sweeps.txt); round 2 added the barrier between the two bursts.
// Same arithmetic, bit for bit.
template <typename T, int NV, int MAXT, bool KC>
__global__ __launch_bounds__(MAXT) void deflate_rows_kernel(
    T* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ t,
    const double* __restrict__ wA, const double* __restrict__ wB, double* __restrict__ ssq_part) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  double* sA = lds;
  double* sB = lds + ((A + 1) & ~1);
  stage_loadings(sA, sB, wA, wB, A, B);
  constexpr int V = VecOf<T>::N;
  using VT = Pack<T, V>;
  const unsigned P = (unsigned)A * (unsigned)B;
  constexpr unsigned stride = (unsigned)MAXT * V;
  const unsigned c0 = threadIdx.x * V;
  const KronWalk w0(c0, stride, B);
  // KC: the stride is a multiple of B, so a lane's vectors all have the same k = c % B: its wB entries
  // stay in registers and j advances by stride / B (no LDS read per element, no index walk)
  const int dj = KC ? (int)(stride / (unsigned)B) : 0;
  double wbv[V];
#pragma unroll
  for (int e = 0; e < V; ++e) wbv[e] = KC ? sB[(w0.k + e < B) ? w0.k + e : 0] : 0.0;
  double ssq = 0.0;
  for (int64_t row = blockIdx.x; row < I; row += gridDim.x) {
    T* __restrict__ xr = X + row * (int64_t)P;
    const double ti = t[row];
    VT x[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const unsigned c = c0 + n * stride;
      x[n] = ld_stream(reinterpret_cast<const VT*>(xr + ((c < P) ? c : 0)));        // clamped: no branch around the load
    }
    // every wavefront's loads have landed before any of them stores: the row goes to HBM as one read burst and one
    // write burst instead of 16 interleaved streams (5.6 -> 5.9 TB/s for this access pattern, tools/exp/rowexp.hip)
    __syncthreads();
    KronWalk w = w0;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const unsigned c = c0 + n * stride;
      if (c < P) {
        const double tw = ti * sA[KC ? w0.j + n * dj : w.j];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T nv = (T)fma(-tw, KC ? wbv[e] : sB[w.k + e], (double)x[n].e[e]);
          x[n].e[e] = nv;
          const double d = (nv == nv) ? (double)nv : 0.0;
          ssq = fma(d, d, ssq);
        }
        st_stream(reinterpret_cast<VT*>(xr + c), x[n]);
      }
      if (!KC) w.next();
    }
  }
  if (ssq_part) {
    const double sblk = block_sum(ssq, red);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = sblk;
  }
}

// ------------------------------------------------------------------------------------------
// center: X[i,c] -= mean[c]; per-row observation counts; sum of squares     wavefront per row
// ------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(kSweepThreads) void center_kernel(
    T* __restrict__ X, int64_t I, int64_t P, const double* __restrict__ mean,
    double* __restrict__ rowcnt, double* __restrict__ ssq_part) {
  __shared__ double red[16];
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (kSweepThreads / kWave);
  const int64_t step = (int64_t)kWave * V;
  double ssq = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * (kSweepThreads / kWave) + (threadIdx.x >> 6); row < I; row += nwaves) {
    T* __restrict__ xr = X + row * P;
    double cnt = 0.0;
    for (int64_t c = (int64_t)lane * V; c < P; c += step) {
      VT x = ld_stream(reinterpret_cast<const VT*>(xr + c));
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const T nv = (T)((double)x.e[e] - mean[c + e]);
        x.e[e] = nv;
        const bool obs = (nv == nv);
        cnt += obs ? 1.0 : 0.0;
        const double d = obs ? (double)nv : 0.0;
        ssq = fma(d, d, ssq);
      }
      st_stream(reinterpret_cast<VT*>(xr + c), x);
    }
    cnt = wave_sum(cnt);
    if (rowcnt && lane == 0) rowcnt[row] = cnt;
  }
  if (ssq_part) {
    const double s = block_sum(ssq, red);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------
// score + deflate fused: one workgroup per row, the row stays in registers between the phases
// ------------------------------------------------------------------------------------------
// KC ("k constant"): the workgroup stride is a multiple of B, so every vector a lane owns has the
// same k = c % B (its wB entries live in registers for the whole kernel) and j advances by a uniform
// stride / B.  This is the shape of every power-of-two benchmark configuration.
// 1024-thread variant (rows up to 256 KB): only NR = NV/2 vectors of a lane stay in registers; the
// other NL = NV/2 are parked in LDS between the phases (128 KB of the CU's 160 KB; one workgroup per
// CU either way), which keeps the kernel inside the 128-VGPR budget of 16 waves/CU without spilling.
template <typename T, bool MASKED, bool VEC, int NV, int MAXT, int MODE>
__global__ __launch_bounds__(MAXT) void score_deflate_kernel(
    T* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ wA,
    const double* __restrict__ wB, const double* __restrict__ rowcnt, double* __restrict__ t,
    double* __restrict__ ssq_part) {
  constexpr bool KC = MODE >= 1;     // stride % B == 0
  constexpr bool FULL = MODE == 2;   // and the workgroup covers the row exactly: no column guards
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  constexpr int NL = (MAXT > 256 && NV >= 16) ? NV / 2 : 0;   // vectors per lane parked in LDS
  constexpr int NR = NV - NL;                     // vectors per lane held in registers
  constexpr int G = 4;                            // parked vectors are streamed G loads at a time
  static_assert(NL % G == 0, "parked vectors come in groups of G");
  extern __shared__ double lds[];
  __shared__ double red[2][16];
  __shared__ double red2[16];
  __shared__ VT park[NL > 0 ? NL * MAXT : 1];
  double* sA = lds;
  double* sB = lds + ((A + 1) & ~1);
  stage_loadings(sA, sB, wA, wB, A, B);
  // a row is < 2^31 elements (guarded on the host): 32-bit element offsets from a uniform row base
  const unsigned P = (unsigned)A * (unsigned)B;
  // the 1024-thread variant is always launched with exactly MAXT threads: a compile-time stride
  const unsigned stride = (MAXT > 256 ? (unsigned)MAXT : blockDim.x) * V;
  const unsigned c0 = threadIdx.x * V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const KronWalk w0(c0, stride, B);   // row independent; re-walked per phase (cheaper than 2*NV registers)
  const int dj = KC ? (int)(stride / (unsigned)B) : 0;
  double wbv[V];
#pragma unroll
  for (int e = 0; e < V; ++e) wbv[e] = KC ? sB[(w0.k + e < B) ? w0.k + e : 0] : 0.0;
  double ssq = 0.0;
  int parity = 0;
  for (int64_t row = blockIdx.x; row < I; row += gridDim.x, parity ^= 1) {
    // scalar row base (opaque to loop strength reduction, which otherwise keeps one 64-bit running
    // address per vector in VGPRs): every access is SGPR base + one shared VGPR offset
    const uint64_t rb = reinterpret_cast<uint64_t>(X + row * (int64_t)P);
    // (the builtin returns int: without the uint32_t casts the low half would be sign-extended)
    const uint32_t rb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rb >> 32));
    const uint32_t rb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb);
    T* __restrict__ xr = reinterpret_cast<T*>(((uint64_t)rb_hi << 32) | (uint64_t)rb_lo);
    VT x[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n)
      if (FULL || c0 + n * stride < P) x[n] = ld_stream(reinterpret_cast<const VT*>((xr + (int64_t)n * stride) + c0));
    double acc = 0.0;
    {
      KronWalk w = w0;
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        if (KC) {
          int jn = w0.j + n * dj;
          asm volatile("" : "+v"(jn));             // keep the sA read inside the row loop (no hoisting into 2*NV registers)
          if (FULL || c0 + n * stride < P) acc = fma(sA[jn], dot_pack<T, V, MASKED>(x[n], wbv), acc);
        } else {
          if (c0 + n * stride < P) acc = fma(sA[w.j], dot_pack<T, V, MASKED>(x[n], sB + w.k), acc);
          w.next();
        }
        if (NV > 4) __builtin_amdgcn_sched_barrier(0);   // keep live temporaries low: the row owns the VGPRs
      }
      // parked half: G loads in flight, scored, then written to this lane's LDS slots
#pragma unroll
      for (int n0 = NR; n0 < NV; n0 += G) {
        VT tmp[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (FULL || c0 + (n0 + g) * stride < P)
            tmp[g] = ld_stream(reinterpret_cast<const VT*>((xr + (int64_t)(n0 + g) * stride) + c0));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int n = n0 + g;
          if (KC) {
            int jn = w0.j + n * dj;
            asm volatile("" : "+v"(jn));
            if (FULL || c0 + n * stride < P) acc = fma(sA[jn], dot_pack<T, V, MASKED>(tmp[g], wbv), acc);
          } else {
            if (c0 + n * stride < P) acc = fma(sA[w.j], dot_pack<T, V, MASKED>(tmp[g], sB + w.k), acc);
            w.next();
          }
          if (FULL || c0 + n * stride < P) park[(n - NR) * MAXT + threadIdx.x] = tmp[g];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[parity][wv] = acc;
    __syncthreads();
    double ti = 0.0;
    for (int w = 0; w < nw; ++w) ti += red[parity][w];
    if (MASKED) ti = ti / rowcnt[row] * (double)P;
    if (threadIdx.x == 0) t[row] = ti;
    if (sizeof(T) == 4) {
      // make the stored row opaque here: otherwise the f64 conversions of phase 1 are kept alive
      // across the barrier for reuse (2 extra VGPRs per element)
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int e = 0; e < V; ++e) asm volatile("" : "+v"(x[n].e[e]));
    }
    {
      KronWalk w = w0;
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        int jn = KC ? w0.j + n * dj : w.j;
        if (KC) asm volatile("" : "+v"(jn));
        if (FULL || c0 + n * stride < P) {
          VT xv;
          if (n < NR) xv = x[n < NR ? n : 0];
          else xv = park[(n - NR) * MAXT + threadIdx.x];   // this lane's own slot: no barrier needed
          const double tw = ti * sA[jn];
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const T nv = (T)fma(-tw, KC ? wbv[e] : sB[w.k + e], (double)xv.e[e]);
            xv.e[e] = nv;
            const double d = (nv == nv) ? (double)nv : 0.0;
            ssq = fma(d, d, ssq);
          }
          st_stream(reinterpret_cast<VT*>((xr + (int64_t)n * stride) + c0), xv);
        }
        if (!KC) w.next();
        if (NV > 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (ssq_part) {
    const double s = block_sum(ssq, red2);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = s;
  }
}


// ------------------------------------------------------------------------------------------
// short rows (P <= 64 * V * 4 elements, e.g. the I x 512 matrix block of a coupled fit, small tensors):
// a wavefront owns RB = 8 / NVL WHOLE rows at a time
// ------------------------------------------------------------------------------------------
// With one row per wavefront a short row keeps only NVL (1-2) 16-byte loads per lane in flight: 2.2-2.7 TB/s at
// 65536 x 512.  Here a wavefront issues the loads of RB consecutive rows (8 per lane in all) before it touches any
// of them; every lane keeps the loadings of ITS columns in registers (no LDS, no index walk), the row sums are
// butterfly wave sums, and the rows never leave the registers between score and deflation (no workgroup barrier).
// OP 0: score (GRAM: + partial sums of Y^T t)   1: deflate with the given t   2: score, then deflate with it.
// Per lane the arithmetic and the column assignment (c = lane * V + n * 64 * V) are those of score_kernel /
// deflate_kernel, bit for bit.
template <typename T, bool MASKED, bool GRAM, int NVL, int OP>
__global__ __launch_bounds__(kSweepThreads) void rows_narrow_kernel(
    T* __restrict__ X, int64_t I, int A, int B, const double* __restrict__ wA, const double* __restrict__ wB,
    const double* __restrict__ rowcnt, double* __restrict__ t, const double* __restrict__ Y, int ldy, int M,
    double* __restrict__ qpart, double* __restrict__ ssq_part) {
  __shared__ double qs[kSweepThreads / kWave][kWave];
  __shared__ double red[16];
  constexpr int V = VecOf<T>::N;
  constexpr int RB = 8 / NVL;
  using VT = Pack<T, V>;
  const int lane = threadIdx.x & 63;
  const int P = A * B;
  double wa[NVL], wb[NVL][V];
  bool ok[NVL];
#pragma unroll
  for (int n = 0; n < NVL; ++n) {
    const int c = lane * V + n * 64 * V;
    ok[n] = c < P;
    const int cs = ok[n] ? c : 0;
    wa[n] = wA[cs / B];                                  // B % V == 0: one j for the whole vector
#pragma unroll
    for (int e = 0; e < V; ++e) wb[n][e] = wB[cs % B + e];
  }
  const int64_t nwaves = (int64_t)gridDim.x * (kSweepThreads / kWave);
  double qacc = 0.0, ssq = 0.0;
  for (int64_t row0 = ((int64_t)blockIdx.x * (kSweepThreads / kWave) + (threadIdx.x >> 6)) * RB; row0 < I; row0 += nwaves * RB) {
    VT x[RB][NVL];
    double yv[RB], tin[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int64_t row = (row0 + r < I) ? row0 + r : I - 1;          // clamped: no branch around the loads
#pragma unroll
      for (int n = 0; n < NVL; ++n) x[r][n] = ld_stream(reinterpret_cast<const VT*>(X + row * P + (ok[n] ? lane * V + n * 64 * V : 0)));
      if (GRAM) yv[r] = Y[row * ldy + ((lane < M) ? lane : 0)];
      if (OP == 1) tin[r] = t[row];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < I;                                 // uniform across the wavefront
      double ti = (OP == 1) ? tin[r] : 0.0;
      if (OP != 1) {
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < NVL; ++n)
          if (ok[n]) acc = fma(wa[n], dot_pack<T, V, MASKED>(x[r][n], wb[n]), acc);
        acc = wave_sum(acc);                                          // identical bits in every lane
        ti = MASKED ? acc / rowcnt[live ? row0 + r : I - 1] * (double)P : acc;
        if (live && lane == 0) t[row0 + r] = ti;
        if (GRAM && live) qacc = fma(ti, yv[r], qacc);
      }
      if (OP != 0 && live) {
#pragma unroll
        for (int n = 0; n < NVL; ++n)
          if (ok[n]) {
            const double tw = ti * wa[n];
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const T nv = (T)fma(-tw, wb[n][e], (double)x[r][n].e[e]);
              x[r][n].e[e] = nv;
              const double d = (nv == nv) ? (double)nv : 0.0;        // NaN (missing) stays NaN, skipped in the norm
              ssq = fma(d, d, ssq);
            }
            st_stream(reinterpret_cast<VT*>(X + (row0 + r) * P + lane * V + n * 64 * V), x[r][n]);
          }
      }
    }
  }
  if (GRAM) {
    qs[threadIdx.x >> 6][lane] = qacc;
    __syncthreads();
    if (threadIdx.x < M) {
      double tot = 0.0;
#pragma unroll
      for (int wv = 0; wv < kSweepThreads / kWave; ++wv) tot += qs[wv][threadIdx.x];
      qpart[(int64_t)blockIdx.x * M + threadIdx.x] = tot;
    }
  }
  if (OP != 0 && ssq_part) {
    const double sblk = block_sum(ssq, red);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = sblk;
  }
}

// vectors per lane per row of the short-row form (0: the row is too long or the shape is not a vector shape)
template <typename T>
static int narrow_nvl(const T* X, int A, int B) {
#ifndef CMTFPLS_ROWS_NARROW
#define CMTFPLS_ROWS_NARROW 1
#endif
  constexpr int V = VecOf<T>::N;
  if (!CMTFPLS_ROWS_NARROW || (B % V) != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0) return 0;
  const int64_t P = (int64_t)A * B;
  if (P <= 64 * V) return 1;
  if (P <= 2 * 64 * V) return 2;
  if (P <= 4 * 64 * V) return 4;
  return 0;
}

template <typename T, bool MASKED, bool GRAM, int OP>
static void launch_rows_narrow(int nvl, hipStream_t st, T* X, int64_t I, int A, int B, const double* wA, const double* wB,
                               const double* rowcnt, double* t, const double* Y, int ldy, int M, double* qpart, double* ssq_part) {
  const dim3 g(kSweepBlocks), b(kSweepThreads);
  if (nvl == 1) hipLaunchKernelGGL((rows_narrow_kernel<T, MASKED, GRAM, 1, OP>), g, b, 0, st, X, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart, ssq_part);
  else if (nvl == 2) hipLaunchKernelGGL((rows_narrow_kernel<T, MASKED, GRAM, 2, OP>), g, b, 0, st, X, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart, ssq_part);
  else hipLaunchKernelGGL((rows_narrow_kernel<T, MASKED, GRAM, 4, OP>), g, b, 0, st, X, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart, ssq_part);
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
template <typename T>
static bool vec_ok(const T* X, int B) {
  return (B % VecOf<T>::N == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
}
static bool shape_ok(int64_t I, int A, int B) { return I > 0 && A > 0 && B > 0; }
static constexpr size_t kMaxLoadingsLds = 96 * 1024;

// Y != nullptr: also writes the kSweepBlocks x M partial sums of Y^T t into qpart
template <typename T>
static int run_score(const T* X, int64_t I, int A, int B, const double* wA, const double* wB,
                     const double* rowcnt, double* t, hipStream_t st,
                     const double* Y = nullptr, int ldy = 0, int M = 0, double* qpart = nullptr, bool few_rows_ok = false) {
  if (!X || !wA || !wB || !t || !shape_ok(I, A, B)) { set_error("score: bad argument"); return CMTFPLS_EINVAL; }
  const bool gram = Y != nullptr;
  if (gram && (!qpart || M <= 0 || ldy < M)) { set_error("score_gram: bad argument"); return CMTFPLS_EINVAL; }
  if (gram && M > kWave) { set_error("score_gram: more than 64 responses; use score + gram_tn"); return CMTFPLS_EUNSUPPORTED; }
  if (const int nvl = narrow_nvl(X, A, B)) {            // short rows: a wavefront owns several whole rows at a time
    T* Xm = const_cast<T*>(X);                           // OP 0 does not write X
    const bool msk = rowcnt != nullptr;
    if (gram) { if (msk) launch_rows_narrow<T, true, true, 0>(nvl, st, Xm, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart, nullptr);
                else launch_rows_narrow<T, false, true, 0>(nvl, st, Xm, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart, nullptr); }
    else      { if (msk) launch_rows_narrow<T, true, false, 0>(nvl, st, Xm, I, A, B, wA, wB, rowcnt, t, nullptr, 0, 0, nullptr, nullptr);
                else launch_rows_narrow<T, false, false, 0>(nvl, st, Xm, I, A, B, wA, wB, rowcnt, t, nullptr, 0, 0, nullptr, nullptr); }
    return check_launch("score");
  }
  size_t lds = loadings_lds_bytes(A, B);
  const bool gl = lds > kMaxLoadingsLds;               // loadings longer than the LDS: read them through L2
  if (gl) lds = 0;
  const bool v = vec_ok(X, B), m = rowcnt != nullptr;
  // a few long rows, one 1024-thread workgroup per row: ONLY for the rows of a cross-covariance S (cmtfpls_score_s_f64), whose row
  // count is the number of responses -- the score of a SAMPLE (cmtfpls_score_*) must not depend on how many samples are
  // passed with it, and this kernel sums in another order than score_kernel
  if (few_rows_ok && !gram && v && !gl && I <= 64 && (int64_t)A * B >= 8192) {
    if (m) hipLaunchKernelGGL((score_fewrows_kernel<T, true>), dim3((unsigned)I), dim3(1024), lds, st, X, A, B, wA, wB, rowcnt, t);
    else hipLaunchKernelGGL((score_fewrows_kernel<T, false>), dim3((unsigned)I), dim3(1024), lds, st, X, A, B, wA, wB, rowcnt, t);
    return check_launch("score");
  }
  const dim3 g(kSweepBlocks), b(kSweepThreads);
#define LAUNCH(MS, V, G) do { if (gl) hipLaunchKernelGGL((score_kernel<T, MS, V, G, true>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart); \
    else hipLaunchKernelGGL((score_kernel<T, MS, V, G, false>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, Y, ldy, M, qpart); } while (0)
  if (gram) { if (m && v) LAUNCH(true, true, true); else if (m) LAUNCH(true, false, true); else if (v) LAUNCH(false, true, true); else LAUNCH(false, false, true); }
  else      { if (m && v) LAUNCH(true, true, false); else if (m) LAUNCH(true, false, false); else if (v) LAUNCH(false, true, false); else LAUNCH(false, false, false); }
#undef LAUNCH
  return check_launch("score");
}

template <typename T>
static int run_deflate(T* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB,
                       double* ssq_part, hipStream_t st) {
  if (!X || !wA || !wB || !t || !shape_ok(I, A, B)) { set_error("deflate: bad argument"); return CMTFPLS_EINVAL; }
  if (const int nvl = narrow_nvl(X, A, B)) {            // short rows: a wavefront owns several whole rows at a time
    launch_rows_narrow<T, false, false, 1>(nvl, st, X, I, A, B, wA, wB, nullptr, const_cast<double*>(t), nullptr, 0, 0, nullptr, ssq_part);
    return check_launch("deflate");
  }
  const size_t lds = loadings_lds_bytes(A, B);
  const dim3 g(kSweepBlocks), b(kSweepThreads);
  if (lds > kMaxLoadingsLds) {                         // loadings longer than the LDS: read them through L2
    if (vec_ok(X, B)) hipLaunchKernelGGL((deflate_kernel<T, true, true>), g, b, 0, st, X, I, A, B, t, wA, wB, ssq_part);
    else hipLaunchKernelGGL((deflate_kernel<T, false, true>), g, b, 0, st, X, I, A, B, t, wA, wB, ssq_part);
    return check_launch("deflate");
  }
#ifndef CMTFPLS_DEFLATE_ROWS
#define CMTFPLS_DEFLATE_ROWS 1
#endif
  if (CMTFPLS_DEFLATE_ROWS && vec_ok(X, B)) {
    const int64_t P = (int64_t)A * B;
    constexpr int V = VecOf<T>::N;
    const bool kc = ((1024 * V) % B) == 0;
#ifndef CMTFPLS_DEFLATE_ROWS_PAD
#define CMTFPLS_DEFLATE_ROWS_PAD 81920
#endif
    // (shadows the outer lds) > half of the CU's 160 KB of LDS: ONE workgroup (one row) per CU at a time --
    // with two, the read/write streams of the rows interleave at the HBM and the sweep is 4 % slower
    const size_t lds_w = loadings_lds_bytes(A, B);
    const size_t lds = lds_w + ((lds_w + CMTFPLS_DEFLATE_ROWS_PAD + 1024 <= 160 * 1024) ? CMTFPLS_DEFLATE_ROWS_PAD : 0);
    if (P > (int64_t)256 * V * 4 && P <= (int64_t)1024 * V * 4) {
      if (kc) hipLaunchKernelGGL((deflate_rows_kernel<T, 4, 1024, true>), g, dim3(1024), lds, st, X, I, A, B, t, wA, wB, ssq_part);
      else hipLaunchKernelGGL((deflate_rows_kernel<T, 4, 1024, false>), g, dim3(1024), lds, st, X, I, A, B, t, wA, wB, ssq_part);
      return check_launch("deflate");
    }
    if (P > (int64_t)1024 * V * 4 && P <= (int64_t)1024 * V * 16) {
      if (kc) hipLaunchKernelGGL((deflate_rows_kernel<T, 16, 1024, true>), g, dim3(1024), lds, st, X, I, A, B, t, wA, wB, ssq_part);
      else hipLaunchKernelGGL((deflate_rows_kernel<T, 16, 1024, false>), g, dim3(1024), lds, st, X, I, A, B, t, wA, wB, ssq_part);
      return check_launch("deflate");
    }
  }
  if (vec_ok(X, B)) hipLaunchKernelGGL((deflate_kernel<T, true>), g, b, lds, st, X, I, A, B, t, wA, wB, ssq_part);
  else hipLaunchKernelGGL((deflate_kernel<T, false>), g, b, lds, st, X, I, A, B, t, wA, wB, ssq_part);
  return check_launch("deflate");
}

// Centring with one WORKGROUP per row segment (the layout of deflate_rows_kernel): a lane owns the same NV = 4
// column vectors in every row it visits, so its means live in registers; the segment is one read burst, the
// observation count is a workgroup sum (whose barrier separates the bursts) and the segment is one write burst.
// Rows longer than 1024 * V * 4 elements are cut into nseg segments; workgroup b keeps segment b % nseg (gridDim
// is a multiple of nseg) and the per-row counts of the segments are added with a f64 atomic -- exact and order
// independent because they are integers (rowcnt is zeroed by the host first).  Same arithmetic per element as
// center_kernel.
template <typename T, int MAXT>
__global__ __launch_bounds__(MAXT) void center_rows_kernel(
    T* __restrict__ X, int64_t I, unsigned P, int nseg, const double* __restrict__ mean,
    double* __restrict__ rowcnt, double* __restrict__ ssq_part) {
  __shared__ double red[2][16];
  __shared__ double red2[16];
  constexpr int V = VecOf<T>::N;
  constexpr int NV = 4;
  using VT = Pack<T, V>;
  constexpr unsigned stride = (unsigned)MAXT * V;
  const unsigned Pseg = P / (unsigned)nseg;                 // host: P % (nseg * V) == 0
  const unsigned seg = blockIdx.x % (unsigned)nseg;
  const unsigned c0 = threadIdx.x * V;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double mu[NV][V];
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < V; ++e) { const unsigned c = c0 + n * stride; mu[n][e] = (c < Pseg) ? mean[seg * Pseg + c + e] : 0.0; }
  double ssq = 0.0;
  int parity = 0;
  for (int64_t row = blockIdx.x / nseg; row < I; row += gridDim.x / nseg, parity ^= 1) {
    T* __restrict__ xr = X + row * (int64_t)P + seg * Pseg;
    VT x[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const unsigned c = c0 + n * stride;
      x[n] = ld_stream(reinterpret_cast<const VT*>(xr + ((c < Pseg) ? c : 0)));     // clamped: no branch around the load
    }
    double cnt = 0.0;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const unsigned c = c0 + n * stride;
      if (c < Pseg) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const T nv = (T)((double)x[n].e[e] - mu[n][e]);
          x[n].e[e] = nv;
          const bool obs = (nv == nv);
          cnt += obs ? 1.0 : 0.0;
          const double d = obs ? (double)nv : 0.0;
          ssq = fma(d, d, ssq);
        }
      }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) red[parity][wv] = cnt;
    __syncthreads();                                     // also: every load has landed before the first store
    if (rowcnt && threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < MAXT / 64; ++w) tot += red[parity][w];
      if (nseg == 1) rowcnt[row] = tot;
      else atomicAdd(rowcnt + row, tot);
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const unsigned c = c0 + n * stride;
      if (c < Pseg) st_stream(reinterpret_cast<VT*>(xr + c), x[n]);
    }
  }
  if (ssq_part) {
    const double sblk = block_sum(ssq, red2);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = sblk;
  }
}

template <typename T>
static int run_center(T* X, int64_t I, int64_t P, const double* mean, double* rowcnt, double* ssq_part, hipStream_t st) {
  if (!X || !mean || I <= 0 || P <= 0) { set_error("center: bad argument"); return CMTFPLS_EINVAL; }
  const dim3 g(kSweepBlocks), b(kSweepThreads);
  const bool v = (P % VecOf<T>::N == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  constexpr int Vc = VecOf<T>::N;
  // rows of more than 256 * V * 4 elements: one workgroup per row segment of <= 1024 * V * 4 elements (read burst,
  // barrier, write burst); > half of the LDS as dynamic padding keeps it at ONE segment per CU, as for
  // deflate_rows_kernel
  if (v && P > (int64_t)256 * Vc * 4 && P < ((int64_t)1 << 31)) {
    const int64_t segmax = (int64_t)1024 * Vc * 4;
    int nseg = (int)((P + segmax - 1) / segmax);
    while (nseg <= 64 && ((P % ((int64_t)nseg * Vc)) != 0 || (kSweepBlocks % nseg) != 0)) ++nseg;
    if (nseg <= 64) {
      if (nseg > 1 && rowcnt) {
        const hipError_t e = hipMemsetAsync(rowcnt, 0, (size_t)I * sizeof(double), st);
        if (e != hipSuccess) { set_error("center: memset failed"); return CMTFPLS_EHIP; }
      }
      hipLaunchKernelGGL((center_rows_kernel<T, 1024>), g, dim3(1024), 81920, st, X, I, (unsigned)P, nseg, mean, rowcnt, ssq_part);
      return check_launch("center");
    }
  }
  if (v) hipLaunchKernelGGL((center_kernel<T, true>), g, b, 0, st, X, I, P, mean, rowcnt, ssq_part);
  else hipLaunchKernelGGL((center_kernel<T, false>), g, b, 0, st, X, I, P, mean, rowcnt, ssq_part);
  return check_launch("center");
}

template <typename T, bool MASKED, bool VEC, int KC>
static void launch_sd(int nv, int threads, size_t lds, hipStream_t st, T* X, int64_t I, int A, int B,
                      const double* wA, const double* wB, const double* rowcnt, double* t, double* ssq_part) {
  const dim3 g(kSweepBlocks), b(threads);
  if (threads > 256 && nv == 4) hipLaunchKernelGGL((score_deflate_kernel<T, MASKED, VEC, 4, 1024, KC>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (threads > 256) hipLaunchKernelGGL((score_deflate_kernel<T, MASKED, VEC, 16, 1024, KC>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (nv == 1) hipLaunchKernelGGL((score_deflate_kernel<T, MASKED, VEC, 1, 256, KC>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else hipLaunchKernelGGL((score_deflate_kernel<T, MASKED, VEC, 4, 256, KC>), g, b, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
}

template <typename T>
static int run_score_deflate(T* X, int64_t I, int A, int B, const double* wA, const double* wB,
                             const double* rowcnt, double* t, double* ssq_part, hipStream_t st) {
  if (!X || !wA || !wB || !t || !shape_ok(I, A, B)) { set_error("score_deflate: bad argument"); return CMTFPLS_EINVAL; }
  if (const int nvl = narrow_nvl(X, A, B)) {            // short rows: the rows stay in registers, no workgroup barrier
    if (rowcnt) launch_rows_narrow<T, true, false, 2>(nvl, st, X, I, A, B, wA, wB, rowcnt, t, nullptr, 0, 0, nullptr, ssq_part);
    else launch_rows_narrow<T, false, false, 2>(nvl, st, X, I, A, B, wA, wB, rowcnt, t, nullptr, 0, 0, nullptr, ssq_part);
    return check_launch("score_deflate");
  }
  const size_t lds = loadings_lds_bytes(A, B);
  if (lds > kMaxLoadingsLds) { set_error("score_deflate: loadings exceed LDS"); return CMTFPLS_EUNSUPPORTED; }
  const bool v = vec_ok(X, B), m = rowcnt != nullptr;
  const int V = v ? VecOf<T>::N : 1;
  const int64_t P = (int64_t)A * B;
  // rows of up to 256*V*4 elements: 256 threads, 1 or 4 vectors per lane; up to 1024*V*4: 1024 threads x 4
  // vectors (16 waves per row hide the latency that 4 waves x 16 vectors could not: 5.0 -> 5.9 TB/s);
  // up to 1024*V*16: 1024 threads x 16 vectors, half of them parked in LDS
  int threads = 256, nv = 1;
  if (P > (int64_t)256 * V * 4) { threads = 1024; nv = 4; }
  if (P > (int64_t)1024 * V * 16) { set_error("score_deflate: row does not fit one workgroup; use score + deflate"); return CMTFPLS_EUNSUPPORTED; }
  while ((int64_t)threads * V * nv < P) nv *= 4;   // 256 threads: 1, 4; 1024 threads: 4, 16
  if (threads == 1024 && nv == 16 && lds + (size_t)1024 * 8 * 16 + 1024 > (size_t)160 * 1024) {
    set_error("score_deflate: loadings + the parked half row exceed the LDS; use score + deflate");
    return CMTFPLS_EUNSUPPORTED;
  }
  const bool kc = v && ((threads * V) % B == 0);
  const bool full = kc && ((int64_t)threads * V * nv == P);
  if (m && full) launch_sd<T, true, true, 2>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (m && kc) launch_sd<T, true, true, 1>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (m && v) launch_sd<T, true, true, 0>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (m) launch_sd<T, true, false, 0>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (full) launch_sd<T, false, true, 2>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (kc) launch_sd<T, false, true, 1>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else if (v) launch_sd<T, false, true, 0>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  else launch_sd<T, false, false, 0>(nv, threads, lds, st, X, I, A, B, wA, wB, rowcnt, t, ssq_part);
  return check_launch("score_deflate");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

int cmtfpls_sweep_partials(void) { return kSweepBlocks; }

size_t cmtfpls_mode0_contract_workspace_bytes(int64_t I, int64_t P) {
  if (I <= 0 || P <= 0) return 0;
  // sized for every plan a call may choose (f32 / f64 vectors, the statistics pass's narrower column tiles)
  int rb = 1;
  for (int elem = 4; elem <= 8; elem += 4)
    for (int U = 2; U <= kContractU; U += 2)
      for (int which = 0; which < 2; ++which) {
        const int r = plan_contract(I, P, elem, U, which ? kContractBlocks : kContractBlocksFull).row_blocks;
        if (r > rb) rb = r;
      }
  return (size_t)rb * (size_t)P * sizeof(double) + (size_t)I * sizeof(double);   // + u = Y q of the wide-block form
}
size_t cmtfpls_colstats_workspace_bytes(int64_t I, int64_t P) { return 2 * cmtfpls_mode0_contract_workspace_bytes(I, P); }

int cmtfpls_colstats_f32(const float* X, int64_t I, int64_t P, double* colsum, double* colcnt, void* ws, size_t n, void* s) {
  return run_contract<float, 2>(X, I, P, nullptr, colsum, colcnt, ws, n, (hipStream_t)s);
}
int cmtfpls_colstats_f64(const double* X, int64_t I, int64_t P, double* colsum, double* colcnt, void* ws, size_t n, void* s) {
  return run_contract<double, 2>(X, I, P, nullptr, colsum, colcnt, ws, n, (hipStream_t)s);
}
int cmtfpls_mode0_contract_f32(const float* X, int64_t I, int64_t P, const double* u, double* Z, int masked, void* ws, size_t n, void* s) {
  return masked ? run_contract<float, 1>(X, I, P, u, Z, nullptr, ws, n, (hipStream_t)s)
                : run_contract<float, 0>(X, I, P, u, Z, nullptr, ws, n, (hipStream_t)s);
}
int cmtfpls_mode0_contract_f64(const double* X, int64_t I, int64_t P, const double* u, double* Z, int masked, void* ws, size_t n, void* s) {
  return masked ? run_contract<double, 1>(X, I, P, u, Z, nullptr, ws, n, (hipStream_t)s)
                : run_contract<double, 0>(X, I, P, u, Z, nullptr, ws, n, (hipStream_t)s);
}
int cmtfpls_mode0_contract_yq_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, const double* q,
                                  double* Z, int masked, void* ws, size_t n, void* s) {
  return masked ? run_contract<float, 1>(X, I, P, nullptr, Z, nullptr, ws, n, (hipStream_t)s, Y, ldy, M, q)
                : run_contract<float, 0>(X, I, P, nullptr, Z, nullptr, ws, n, (hipStream_t)s, Y, ldy, M, q);
}
int cmtfpls_mode0_contract_yq_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, const double* q,
                                  double* Z, int masked, void* ws, size_t n, void* s) {
  return masked ? run_contract<double, 1>(X, I, P, nullptr, Z, nullptr, ws, n, (hipStream_t)s, Y, ldy, M, q)
                : run_contract<double, 0>(X, I, P, nullptr, Z, nullptr, ws, n, (hipStream_t)s, Y, ldy, M, q);
}
size_t cmtfpls_deflate_contract_workspace_bytes(int64_t I, int64_t P) {
  if (I <= 0 || P <= 0) return 0;
  const ContractPlan a = plan_contract(I, P, 4, kDcU), b = plan_contract(I, P, 8, kDcU);
  const size_t na = (size_t)a.row_blocks * ((size_t)P + a.col_tiles), nb = (size_t)b.row_blocks * ((size_t)P + b.col_tiles);
  const size_t colform = ((na > nb ? na : nb) + (size_t)I) * sizeof(double);   // + u = Y q of the wide-block form
  // workgroup-per-row-segment form: at most kDcRowsGrid partial rows of P doubles, the ssq partials, u = Y q
  const size_t rowform = ((size_t)kDcRowsGrid * (size_t)P + (size_t)kDcRowsGrid + (size_t)I) * sizeof(double);
  return colform > rowform ? colform : rowform;
}
int cmtfpls_deflate_contract_yq_f32(float* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB,
                                    const double* Y, int ldy, int M, const double* q, double* Z, int masked, double* ssq,
                                    void* ws, size_t n, void* s) {
  return run_deflate_contract<float>(X, I, A, B, t, wA, wB, Y, ldy, M, q, Z, masked, ssq, ws, n, (hipStream_t)s);
}
int cmtfpls_deflate_contract_yq_f64(double* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB,
                                    const double* Y, int ldy, int M, const double* q, double* Z, int masked, double* ssq,
                                    void* ws, size_t n, void* s) {
  return run_deflate_contract<double>(X, I, A, B, t, wA, wB, Y, ldy, M, q, Z, masked, ssq, ws, n, (hipStream_t)s);
}
int cmtfpls_score_gram_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt,
                           double* t, const double* Y, int ldy, int M, double* qpart, void* s) {
  if (!Y) { set_error("score_gram: Y is null"); return CMTFPLS_EINVAL; }
  return run_score<float>(X, I, A, B, wA, wB, rowcnt, t, (hipStream_t)s, Y, ldy, M, qpart);
}
int cmtfpls_score_gram_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt,
                           double* t, const double* Y, int ldy, int M, double* qpart, void* s) {
  if (!Y) { set_error("score_gram: Y is null"); return CMTFPLS_EINVAL; }
  return run_score<double>(X, I, A, B, wA, wB, rowcnt, t, (hipStream_t)s, Y, ldy, M, qpart);
}
int cmtfpls_center_f32(float* X, int64_t I, int64_t P, const double* mean, double* rowcnt, double* ssq_part, void* s) {
  return run_center<float>(X, I, P, mean, rowcnt, ssq_part, (hipStream_t)s);
}
int cmtfpls_center_f64(double* X, int64_t I, int64_t P, const double* mean, double* rowcnt, double* ssq_part, void* s) {
  return run_center<double>(X, I, P, mean, rowcnt, ssq_part, (hipStream_t)s);
}
int cmtfpls_score_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt, double* t, void* s) {
  return run_score<float>(X, I, A, B, wA, wB, rowcnt, t, (hipStream_t)s);
}
int cmtfpls_score_s_f64(const double* S, int M, int A, int B, const double* wA, const double* wB, double* tq, void* s) {
  return run_score<double>(S, M, A, B, wA, wB, nullptr, tq, (hipStream_t)s, nullptr, 0, 0, nullptr, true);
}
int cmtfpls_score_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt, double* t, void* s) {
  return run_score<double>(X, I, A, B, wA, wB, rowcnt, t, (hipStream_t)s);
}
int cmtfpls_deflate_f32(float* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB, double* ssq_part, void* s) {
  return run_deflate<float>(X, I, A, B, t, wA, wB, ssq_part, (hipStream_t)s);
}
int cmtfpls_deflate_f64(double* X, int64_t I, int A, int B, const double* t, const double* wA, const double* wB, double* ssq_part, void* s) {
  return run_deflate<double>(X, I, A, B, t, wA, wB, ssq_part, (hipStream_t)s);
}
int cmtfpls_score_deflate_f32(float* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt, double* t, double* ssq_part, void* s) {
  return run_score_deflate<float>(X, I, A, B, wA, wB, rowcnt, t, ssq_part, (hipStream_t)s);
}
int cmtfpls_score_deflate_f64(double* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* rowcnt, double* t, double* ssq_part, void* s) {
  return run_score_deflate<double>(X, I, A, B, wA, wB, rowcnt, t, ssq_part, (hipStream_t)s);
}

}  // extern "C"
