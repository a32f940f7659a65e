// K8: factors_to_tensor (cmtf_pls/util.py:18-20) as used by X_reconstructed (tpls.py:188-189, cmtf.py:233-237):
//   Xhat[i, c] = sum_r T[i, r] * WA[c / B, r] * WB[c % B, r]  (+ mean[c])
// for a block of rows, written in the storage type of X.  The reference forms T . khatri_rao(...)^T as a dense
// GEMM on the host (8.6 GB of float64 at 65536 x 128 x 128); here the Khatri-Rao operand is never materialised:
// a thread owns V consecutive columns (16 bytes of output), keeps their R loading products in registers and
// loops over the rows of its row block; the score row T[i, :] is workgroup-uniform (scalar loads).
// The kernel is write-bound (I * P * s bytes out, R f64 FMAs per element: 10 FMAs against 4 bytes).
#include "common.hpp"

namespace cmtfpls {

constexpr int kReconMaxR = 16;   // components per pass (register budget); more are accumulated in passes

// VEC false: any shape (B not a multiple of 16 bytes, unaligned output): one element per thread, scalar accesses
template <typename T, int RC, bool VEC>
__global__ __launch_bounds__(kSweepThreads) void recon_kernel(const double* __restrict__ Tm, int ldt, int r0, int R,
                                                             const double* __restrict__ WA, const double* __restrict__ WB, int B,
                                                             const double* __restrict__ mean, T* __restrict__ out, int64_t I, int64_t P,
                                                             int rows_per_block, int accumulate) {
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  const int64_t c = ((int64_t)blockIdx.x * kSweepThreads + threadIdx.x) * V;
  if (c >= P) return;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t i1 = (i0 + rows_per_block < I) ? i0 + rows_per_block : I;
  double w[RC][V], mu[V];
  const int j = (int)(c / B), k = (int)(c % B);          // B % V == 0: one j for the whole vector
#pragma unroll
  for (int r = 0; r < RC; ++r)
#pragma unroll
    for (int e = 0; e < V; ++e) w[r][e] = (r0 + r < R) ? WA[(int64_t)j * R + r0 + r] * WB[(int64_t)(k + e) * R + r0 + r] : 0.0;
#pragma unroll
  for (int e = 0; e < V; ++e) mu[e] = (mean && !accumulate) ? mean[c + e] : 0.0;
  for (int64_t i = i0; i < i1; ++i) {
    const double* __restrict__ trow = Tm + i * ldt + r0;
    double acc[V];
    VT o;
    if (accumulate) {
      o = *reinterpret_cast<const VT*>(out + i * P + c);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = (double)o.e[e];
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = mu[e];
    }
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const double tr = (r0 + r < R) ? trow[r] : 0.0;    // uniform across the workgroup
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = fma(tr, w[r][e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) o.e[e] = (T)acc[e];
    st_stream(reinterpret_cast<VT*>(out + i * P + c), o);
  }
}

template <typename T>
static int run_recon(const double* Tm, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                     const double* mean, T* out, hipStream_t st) {
  if (!Tm || !WA || !WB || !out || I <= 0 || R <= 0 || A <= 0 || B <= 0 || ldt < R) { set_error("recon: bad argument"); return CMTFPLS_EINVAL; }
  const bool vec = (B % VecOf<T>::N) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const int V = vec ? VecOf<T>::N : 1;
  const int64_t P = (int64_t)A * B;
  const int col_tiles = (int)((P / V + kSweepThreads - 1) / kSweepThreads);
  int64_t want = (2048 + col_tiles - 1) / col_tiles;
  int64_t rpb = (I + want - 1) / want;
  if (rpb < 8) rpb = 8;
  const int row_blocks = (int)((I + rpb - 1) / rpb);
  const dim3 grid(col_tiles, row_blocks), block(kSweepThreads);
#define RK(RCC, VV) hipLaunchKernelGGL((recon_kernel<T, RCC, VV>), grid, block, 0, st, Tm, ldt, r0, R, WA, WB, B, mean, out, I, P, (int)rpb, acc)
  for (int r0 = 0; r0 < R; r0 += kReconMaxR) {
    const int rc = (R - r0 < kReconMaxR) ? R - r0 : kReconMaxR;
    const int acc = r0 > 0;
    if (vec) { if (rc <= 4) RK(4, true); else if (rc <= 8) RK(8, true); else if (rc <= 12) RK(12, true); else RK(16, true); }
    else     { if (rc <= 4) RK(4, false); else if (rc <= 8) RK(8, false); else if (rc <= 12) RK(12, false); else RK(16, false); }
  }
#undef RK
  return check_launch("recon");
}

// calcR2X (util.py:7-15) against the reconstruction WITHOUT materialising it: for the rows of one block
//   part[blk][0] = sum over finite x of (xhat - x)^2,   part[blk][1] = sum over finite x of x^2,
// x = X[i, c] - mean[c] (the centred original, tpls.py:115-117), xhat = sum_r T[i, r] WA[c / B, r] WB[c % B, r].
// One read of X; same thread layout as recon_kernel.
template <typename T, int RC, bool VEC>
__global__ __launch_bounds__(kSweepThreads) void recon_r2_kernel(const T* __restrict__ X, const double* __restrict__ Tm, int ldt, int R,
                                                                const double* __restrict__ WA, const double* __restrict__ WB, int B,
                                                                const double* __restrict__ mean, int64_t I, int64_t P, int rows_per_block,
                                                                double* __restrict__ part) {
  __shared__ double red[16];
  constexpr int V = VEC ? VecOf<T>::N : 1;
  using VT = Pack<T, V>;
  const int64_t c = ((int64_t)blockIdx.x * kSweepThreads + threadIdx.x) * V;
  const bool live = c < P;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t i1 = (i0 + rows_per_block < I) ? i0 + rows_per_block : I;
  double w[RC][V], mu[V];
  const int64_t cs = live ? c : 0;
  const int j = (int)(cs / B), k = (int)(cs % B);
#pragma unroll
  for (int r = 0; r < RC; ++r)
#pragma unroll
    for (int e = 0; e < V; ++e) w[r][e] = (r < R) ? WA[(int64_t)j * R + r] * WB[(int64_t)(k + e) * R + r] : 0.0;
#pragma unroll
  for (int e = 0; e < V; ++e) mu[e] = mean ? mean[cs + e] : 0.0;
  double res = 0.0, ssq = 0.0;
  if (live) {
    for (int64_t i = i0; i < i1; ++i) {
      const double* __restrict__ trow = Tm + i * ldt;
      const VT x = ld_stream(reinterpret_cast<const VT*>(X + i * P + c));
      double acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.0;
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        const double tr = (r < R) ? trow[r] : 0.0;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = fma(tr, w[r][e], acc[e]);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const double xc = (double)x.e[e] - mu[e];
        const bool fin = isfinite(xc);                      // np.isfinite(X) mask of util.py:11-13
        const double d = fin ? acc[e] - xc : 0.0;
        res = fma(d, d, res);
        ssq = fma(fin ? xc : 0.0, fin ? xc : 0.0, ssq);
      }
    }
  }
  const double r1 = block_sum(res, red);
  __syncthreads();
  const double r2 = block_sum(ssq, red);
  if (threadIdx.x == 0) {
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * blk] = r1;
    part[2 * blk + 1] = r2;
  }
}

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st);

static void recon_r2_plan(int64_t I, int64_t P, int V, int* col_tiles, int* row_blocks, int64_t* rpb) {
  *col_tiles = (int)(((P + V - 1) / V + kSweepThreads - 1) / kSweepThreads);      // (>= 1 for P >= 1: a block of fewer columns than a
  if (*col_tiles < 1) *col_tiles = 1;                                             //  vector made this 0 and the next line divide by it)
  int64_t want = (2048 + *col_tiles - 1) / *col_tiles;
  *rpb = (I + want - 1) / want;
  if (*rpb < 8) *rpb = 8;
  *row_blocks = (int)((I + *rpb - 1) / *rpb);
}

template <typename T>
static int run_recon_r2(const T* X, const double* Tm, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                        const double* mean, double* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!X || !Tm || !WA || !WB || !out || I <= 0 || R <= 0 || A <= 0 || B <= 0 || ldt < R) { set_error("recon_r2: bad argument"); return CMTFPLS_EINVAL; }
  if (R > kReconMaxR) { set_error("recon_r2: more than 16 components"); return CMTFPLS_EUNSUPPORTED; }
  const bool vec = (B % VecOf<T>::N) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  const int V = vec ? VecOf<T>::N : 1;
  const int64_t P = (int64_t)A * B;
  int col_tiles, row_blocks;
  int64_t rpb;
  recon_r2_plan(I, P, V, &col_tiles, &row_blocks, &rpb);
  const size_t nblk = (size_t)col_tiles * row_blocks;
  if (!ws || ws_bytes < nblk * 2 * sizeof(double)) { set_error("recon_r2: workspace too small"); return CMTFPLS_EWORKSPACE; }
  double* part = static_cast<double*>(ws);
  const dim3 grid(col_tiles, row_blocks), block(kSweepThreads);
#define R2K(RCC, VV) hipLaunchKernelGGL((recon_r2_kernel<T, RCC, VV>), grid, block, 0, st, X, Tm, ldt, R, WA, WB, B, mean, I, P, (int)rpb, part)
  if (vec) { if (R <= 4) R2K(4, true); else if (R <= 8) R2K(8, true); else if (R <= 12) R2K(12, true); else R2K(16, true); }
  else     { if (R <= 4) R2K(4, false); else if (R <= 8) R2K(8, false); else if (R <= 12) R2K(12, false); else R2K(16, false); }
#undef R2K
  launch_reduce_rows(part, (int)nblk, 2, out, st);
  return check_launch("recon_r2");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
int cmtfpls_recon_f32(const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                      const double* mean, float* out, void* stream) {
  return run_recon<float>(T, I, ldt, R, WA, WB, A, B, mean, out, (hipStream_t)stream);
}
int cmtfpls_recon_f64(const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                      const double* mean, double* out, void* stream) {
  return run_recon<double>(T, I, ldt, R, WA, WB, A, B, mean, out, (hipStream_t)stream);
}
size_t cmtfpls_recon_r2_workspace_bytes(int64_t I, int64_t P) {
  if (I <= 0 || P <= 0) return 0;
  size_t most = 0;
  for (int V = 1; V <= 4; V *= 2) {
    int ct, rb;
    int64_t rpb;
    recon_r2_plan(I, P, V, &ct, &rb, &rpb);
    const size_t nb = (size_t)ct * rb * 2 * sizeof(double);
    if (nb > most) most = nb;
  }
  return most;
}
int cmtfpls_recon_r2_f32(const float* X, const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                         const double* mean, double* out, void* ws, size_t ws_bytes, void* stream) {
  return run_recon_r2<float>(X, T, I, ldt, R, WA, WB, A, B, mean, out, ws, ws_bytes, (hipStream_t)stream);
}
int cmtfpls_recon_r2_f64(const double* X, const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                         const double* mean, double* out, void* ws, size_t ws_bytes, void* stream) {
  return run_recon_r2<double>(X, T, I, ldt, R, WA, WB, A, B, mean, out, ws, ws_bytes, (hipStream_t)stream);
}
}
