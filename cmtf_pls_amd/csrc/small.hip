// Small f64 algebra of the NIPALS loop on tall-skinny operands (Y, T, u, q): everything the
// reference does with BLAS level-1/2 calls between the two X sweeps (tpls.py:100-113).  All sums
// are two-stage with a fixed order so results are bit-reproducible run to run and rank to rank.
#include "common.hpp"

namespace cmtfpls {

void launch_reduce_rows(const double* part, int nrows, int64_t P, double* out, hipStream_t st);

constexpr int kSmallBlocks = 128;
constexpr int kMaxGramDim = 64;
constexpr size_t kSmallWsBytes = (size_t)kSmallBlocks * kMaxGramDim * kMaxGramDim * sizeof(double);

// C = A^T B, per-workgroup partials: part[blk][p*b + q]
__global__ __launch_bounds__(256) void gram_tn_kernel(const double* __restrict__ A, int lda, int a,
                                                     const double* __restrict__ B, int ldb, int b,
                                                     int64_t I, double* __restrict__ part, int TR) {
  extern __shared__ double lds[];
  double* As = lds;            // TR x a
  double* Bs = lds + TR * a;   // TR x b
  const int ab = a * b;
  const int64_t rows_per = (I + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = (r0 + rows_per < I) ? r0 + rows_per : I;
  double acc[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) acc[s] = 0.0;
  for (int64_t rb = r0; rb < r1; rb += TR) {
    const int nr = (int)((r1 - rb < TR) ? r1 - rb : TR);
    if (lda == a) {                                          // contiguous rows (Y, a score vector): no index arithmetic
      for (int idx = threadIdx.x; idx < nr * a; idx += 256) As[idx] = A[rb * a + idx];
    } else {
      for (int idx = threadIdx.x; idx < nr * a; idx += 256) As[idx] = A[(rb + idx / a) * lda + idx % a];
    }
    if (ldb == b) {
      for (int idx = threadIdx.x; idx < nr * b; idx += 256) Bs[idx] = B[rb * b + idx];
    } else {
      for (int idx = threadIdx.x; idx < nr * b; idx += 256) Bs[idx] = B[(rb + idx / b) * ldb + idx % b];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int o = threadIdx.x + s * 256;
      if (o < ab) {
        const int p = o / b, q = o % b;
        double v = acc[s];
        for (int r = 0; r < nr; ++r) v = fma(As[r * a + p], Bs[r * b + q], v);
        acc[s] = v;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int o = threadIdx.x + s * 256;
    if (o < ab) part[(int64_t)blockIdx.x * ab + o] = acc[s];
  }
}

// C[p*ldc + q] = sum over workgroups of part[blk][p*bc + q]: one wavefront per output element,
// lanes stride over the partials, butterfly sum (fixed order -> bit-reproducible)
__global__ __launch_bounds__(256) void reduce_block_kernel(const double* __restrict__ part, int nblk, int ac, int bc,
                                                          double* __restrict__ C, int ldc) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= ac * bc) return;
  double s = 0.0;
  for (int r = lane; r < nblk; r += 64) s += part[(int64_t)r * ac * bc + o];
  s = wave_sum(s);
  if (lane == 0) C[(int64_t)(o / bc) * ldc + (o % bc)] = s;
}

// q = A^T v for a tall A (I x a, a <= 16 per pass) and a vector v: thread per row, the a
// accumulators stay in registers, one partial row per workgroup.  This is Y.T @ t (tpls.py:100).
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, int lda, int a0, int ac,
                                                    const double* __restrict__ v, int ldv, int64_t I,
                                                    double* __restrict__ part, int a_total) {
  __shared__ double red[4][16];
  double acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < I; i += (int64_t)gridDim.x * 256) {
    const double vi = v[i * ldv];
    const double* ar = A + i * lda + a0;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < ac) acc[e] = fma(ar[e], vi, acc[e]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const double w = wave_sum(acc[e]);
    if (lane == 0) red[wv][e] = w;
  }
  __syncthreads();
  if (threadIdx.x < ac) {
    const int e = threadIdx.x;
    part[(int64_t)blockIdx.x * a_total + a0 + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
  }
}

// u[i] = Y[i,:] . q ;  optional partial of sum (u_old - u)^2
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ Y, int ldy, int M, int64_t I,
                                                    const double* __restrict__ q, double* __restrict__ u,
                                                    const double* __restrict__ u_old, double* __restrict__ part) {
  __shared__ double red[16];
  double d2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < I; i += (int64_t)gridDim.x * 256) {
    const double* yr = Y + i * ldy;
    double s = 0.0;
    for (int m = 0; m < M; ++m) s = fma(yr[m], q[m], s);
    if (u_old) { const double d = u_old[i] - s; d2 = fma(d, d, d2); }
    u[i] = s;
  }
  if (part) {
    const double s = block_sum(d2, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
  }
}

// Y[i,m] -= (T[i,:R] . b) q[m]; partial of ||Y||_F^2 afterwards
__global__ __launch_bounds__(256) void y_deflate_kernel(double* __restrict__ Y, int ldy, int M, int64_t I,
                                                       const double* __restrict__ T, int ldt, int R,
                                                       const double* __restrict__ b, const double* __restrict__ q,
                                                       double* __restrict__ part) {
  __shared__ double red[16];
  double ssq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < I; i += (int64_t)gridDim.x * 256) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) s = fma(T[i * ldt + r], b[r], s);
    double* yr = Y + i * ldy;
    for (int m = 0; m < M; ++m) {
      const double v = yr[m] - s * q[m];
      yr[m] = v;
      ssq = fma(v, v, ssq);
    }
  }
  const double s = block_sum(ssq, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = sum(in[0..n)) : one workgroup, strided partials then a fixed-order tree
__global__ __launch_bounds__(1024) void sum_kernel(const double* __restrict__ in, int64_t n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += in[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// v /= ||v||_2 : one workgroup
__global__ __launch_bounds__(1024) void normalize_kernel(double* __restrict__ v, int64_t n, double* __restrict__ nrm) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s = fma(v[i], v[i], s);
  s = sqrt(block_sum(s, red));
  for (int64_t i = threadIdx.x; i < n; i += 1024) v[i] = v[i] / s;
  if (nrm && threadIdx.x == 0) nrm[0] = s;
}

// out = in / ||in||_2 : one workgroup (normalize_kernel with the copy folded in: the loading of a matrix block, tpls.py:84-90 for a vector Z)
__global__ __launch_bounds__(1024) void normalize_to_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t n) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s = fma(in[i], in[i], s);
  s = sqrt(block_sum(s, red));
  for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = in[i] / s;
}

// A MATRIX block's share of one inner iteration on S in ONE workgroup (cmtf.py:91-119 re-associated; three launches before):
//   Z = sum_m q[m] S[m,:]  (masked: x n_samples / colcnt, missingvals.py:17-19),   wB = Z / |Z|  (tpls.py:84-90 for a vector Z),
//   tq[m] = S2[m,:] . wB   (the block's Y^T t; S2 = S without missing values).
// The loading stays in LDS between the steps (P <= 8192 doubles); wavefront w forms tq[w], tq[w + 16], ...
__global__ __launch_bounds__(1024) void s_vector_block_kernel(const double* __restrict__ S, const double* __restrict__ S2,
                                                            const double* __restrict__ colcnt, double n_samples, int M, int P,
                                                            const double* __restrict__ q, double* __restrict__ Z,
                                                            double* __restrict__ wB, double* __restrict__ tq) {
  extern __shared__ double wl[];
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double ssq = 0.0;
  for (int c = threadIdx.x; c < P; c += 1024) {
    double z = 0.0;
#pragma unroll 8
    for (int m = 0; m < M; ++m) z = fma(q[m], S[(int64_t)m * P + c], z);
    if (colcnt) z = (colcnt[c] > 0.0) ? z / colcnt[c] * n_samples : 0.0;
    Z[c] = z;
    wl[c] = z;
    ssq = fma(z, z, ssq);
  }
  const double nrm = sqrt(block_sum(ssq, red));
  for (int c = threadIdx.x; c < P; c += 1024) {
    const double w = wl[c] / nrm;
    wl[c] = w;
    wB[c] = w;
  }
  __syncthreads();
  for (int m = wv; m < M; m += 16) {
    double acc = 0.0;
    for (int c = lane; c < P; c += 64) acc = fma(S2[(int64_t)m * P + c], wl[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) tq[m] = acc;
  }
}

// The Y-side update of one NIPALS iteration in one workgroup (tpls.py:100-103):
//   q <- sum over workgroups of qpart (the score kernel's partial sums of Y^T t)      [qpart != null]
//   q <- q / |q|                                                                      [normalize]
//   du2 <- (q - q_prev)^T G (q - q_prev)  =  |Y q - Y q_prev|^2  with G = Y^T Y       [G != null]
// Every sum has a fixed order (bit-reproducible).  M <= 64.
__global__ __launch_bounds__(1024) void q_update_kernel(const double* __restrict__ qpart, int nblk, int M,
                                                       double* __restrict__ q, int normalize,
                                                       const double* __restrict__ G, const double* __restrict__ q_prev,
                                                       double* __restrict__ du2) {
  __shared__ double part[64][17];          // [group][m] for Mp = 16; reshaped below for wider M
  __shared__ double qs[64];
  __shared__ double dq[64];
  __shared__ double red[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int Mp = (M <= 16) ? 16 : (M <= 32) ? 32 : 64;
  const int ngrp = 1024 / Mp;               // 64, 32 or 16 groups of Mp lanes
  const int m = tid & (Mp - 1), grp = tid / Mp;
  double* pflat = &part[0][0];              // ngrp x (Mp + 1) <= 64 x 17 = 16 x 65 + ... fits 1088 doubles
  if (qpart) {
    // batches of 16 independent loads (clamped address, masked value): one memory latency per batch
    double sacc = 0.0;
    const int msafe = (m < M) ? m : 0;
    for (int b0 = grp; b0 < nblk; b0 += 16 * ngrp) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int b = b0 + k * ngrp;
        v[k] = qpart[(int64_t)((b < nblk) ? b : nblk - 1) * M + msafe];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) sacc += (b0 + k * ngrp < nblk) ? v[k] : 0.0;
    }
    if (m >= M) sacc = 0.0;
    pflat[grp * (Mp + 1) + m] = sacc;
    __syncthreads();
    if (tid < M) {
      double tot = 0.0;
      for (int g = 0; g < ngrp; ++g) tot += pflat[g * (Mp + 1) + tid];
      qs[tid] = tot;
    }
  } else if (tid < M) {
    qs[tid] = q[tid];
  }
  __syncthreads();
  if (tid < 64) {
    // wavefront 0: norm by a butterfly over the (zero padded) 64 lanes
    double v = (lane < M) ? qs[lane] : 0.0;
    if (normalize) {
      const double nrm = sqrt(wave_sum(v * v));
      v = v / nrm;
    }
    if (lane < M) {
      q[lane] = v;
      if (G) dq[lane] = v - q_prev[lane];
    }
  }
  if (G) {
    __syncthreads();
    double sacc = 0.0;
    for (int idx = tid; idx < M * M; idx += 1024) {
      const int i = idx / M, j = idx - i * M;
      sacc = fma(dq[i] * G[idx], dq[j], sacc);
    }
    sacc = block_sum(sacc, red);
    if (tid == 0) du2[0] = sacc;
  }
}

// Down-date of the cross-covariance S = Y^T X_(0) (M x P) across one deflation (tpls.py:109,113):
//   X+ = X - t w^T,  Y+ = Y - yhat q^T   =>   S+ = S - (Y^T t) w^T - q (X+^T yhat)^T
// with w[c] = wA[c / B] wB[c % B] formed on the fly.  One thread per column, M rows each.
__global__ __launch_bounds__(256) void s_downdate_kernel(double* __restrict__ S, int M, int64_t P, int B,
                                                        const double* __restrict__ ya, const double* __restrict__ wA,
                                                        const double* __restrict__ wB, const double* __restrict__ q,
                                                        const double* __restrict__ v) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= P) return;
  const double w = wA[c / B] * wB[c % B];
  const double vc = v[c];
  for (int m = 0; m < M; ++m) {
    double sv = S[(int64_t)m * P + c];
    sv = fma(-ya[m], w, sv);
    sv = fma(-q[m], vc, sv);
    S[(int64_t)m * P + c] = sv;
  }
}

// Z[c] = sum_m q[m] S[m, c]: np.einsum("i...,i->...", X, u) (tpls.py:83) re-associated through S = Y^T X_(0) with u = Y q.
// S has M <= 64 rows and lives in L2: one thread per column, the M rows in index order (one fixed fma chain per column), no
// partial rows and no second kernel -- the general contraction (sweeps.hip) splits rows over workgroups and adds a
// reduce_rows launch, one launch of pure latency too many in an iteration that is ~14 of them.
__global__ __launch_bounds__(256) void s_contract_kernel(const double* __restrict__ S, int M, int64_t P, const double* __restrict__ q,
                                                        double* __restrict__ Z) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= P) return;
  double acc = 0.0;
#pragma unroll 8
  for (int m = 0; m < M; ++m) acc = fma(q[m], S[(int64_t)m * P + c], acc);
  Z[c] = acc;
}

// y[i] -= a[0] * (x ? x[i] : 1): the two rank-one corrections of the cross-covariance loop on an UNCENTRED X (round 3):
//   X_c w = X w - (mean^T w) 1   (a scalar shift of the I scores),   X_c^T yhat = X^T yhat - (1^T yhat) mean   (P entries).
__global__ __launch_bounds__(256) void axpy_scalar_kernel(double* __restrict__ y, int64_t n, const double* __restrict__ a,
                                                         const double* __restrict__ x) {
  const double av = a[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = fma(-av, x ? x[i] : 1.0, y[i]);
}

// v[c] -= sum_{j < k} coef[j] * WA[(c / B) * ld + j] * WB[(c % B) * ld + j]: a rank-k correction in Khatri-Rao form, the
// Khatri-Rao product never materialised.  Used by the xcov algorithm when X is NOT deflated in place: with
// X_a = X_0 - sum_{j<a} t_j w_j^T the contraction the S down-date needs is X_{a+1}^T yhat = X_0^T yhat - sum_{j<=a} w_j (t_j^T yhat).
__global__ __launch_bounds__(256) void kr_axpy_kernel(double* __restrict__ v, int64_t P, int B, const double* __restrict__ WA,
                                                     const double* __restrict__ WB, int ld, int k, const double* __restrict__ coef) {
  __shared__ double cs[64];
  for (int j = threadIdx.x; j < k; j += 256) cs[j] = coef[j];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= P) return;
  const double* __restrict__ wa = WA + (c / B) * ld;
  const double* __restrict__ wb = WB + (c % B) * ld;
  double s = 0.0;
  for (int j = 0; j < k; ++j) s = fma(cs[j] * wa[j], wb[j], s);
  v[c] -= s;
}

__global__ __launch_bounds__(256) void colscale_kernel(double* __restrict__ Z, int64_t P, const double* __restrict__ cnt, double n_samples) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < P) Z[c] = (cnt[c] > 0.0) ? Z[c] / cnt[c] * n_samples : 0.0;
}

__global__ __launch_bounds__(256) void scores_mean_kernel(const double* __restrict__ Ts, int nb, int64_t I, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= I) return;
  double s = Ts[i];
  for (int b = 1; b < nb; ++b) s += Ts[(int64_t)b * I + i];
  out[i] = s / (double)nb;
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_small_workspace_bytes(void) { return kSmallWsBytes; }

int cmtfpls_gram_tn_f64(const double* A, int lda, int a, const double* B, int ldb, int b, int64_t I,
                        double* C, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || a <= 0 || b <= 0 || lda < a || ldb < b || I <= 0) {
    set_error("gram_tn: bad argument");
    return CMTFPLS_EINVAL;
  }
  if (!ws || ws_bytes < kSmallWsBytes) { set_error("gram_tn: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = static_cast<double*>(ws);
  if (b == 1 && (size_t)kSmallBlocks * a * sizeof(double) <= kSmallWsBytes) {
    // A^T v: register accumulators, 16 columns of A per pass
    for (int a0 = 0; a0 < a; a0 += 16) {
      const int ac = (a - a0 < 16) ? a - a0 : 16;
      hipLaunchKernelGGL(gemv_t_kernel, dim3(kSmallBlocks), dim3(256), 0, st, A, lda, a0, ac, B, ldb, I, part, a);
    }
    hipLaunchKernelGGL(reduce_block_kernel, dim3((a + 3) / 4), dim3(256), 0, st, part, kSmallBlocks, a, 1, C, 1);
    return check_launch("gram_tn");
  }
  // wide operands (e.g. Y with hundreds of responses) are tiled into <= 64 x 64 output blocks
  for (int a0 = 0; a0 < a; a0 += kMaxGramDim)
    for (int b0 = 0; b0 < b; b0 += kMaxGramDim) {
      const int ac = (a - a0 < kMaxGramDim) ? a - a0 : kMaxGramDim;
      const int bc = (b - b0 < kMaxGramDim) ? b - b0 : kMaxGramDim;
      // rows staged per barrier pair: as many as 32 KB of LDS hold (32 rows of a 16 x 16 Gram left the kernel waiting on its
      // barriers: 31-39 us for 65536 rows)
      const int tr = (ac + bc <= 32) ? 128 : (ac + bc <= 64) ? 64 : 32;
      const size_t lds = (size_t)tr * (ac + bc) * sizeof(double);
      hipLaunchKernelGGL(gram_tn_kernel, dim3(kSmallBlocks), dim3(256), lds, st, A + a0, lda, ac, B + b0, ldb, bc, I, part, tr);
      hipLaunchKernelGGL(reduce_block_kernel, dim3((ac * bc + 3) / 4), dim3(256), 0, st, part, kSmallBlocks, ac, bc,
                         C + (size_t)a0 * b + b0, b);
    }
  return check_launch("gram_tn");
}

int cmtfpls_rowdot_f64(const double* Y, int ldy, int M, int64_t I, const double* q, double* u,
                       const double* u_old, double* du2, void* ws, size_t ws_bytes, void* stream) {
  if (!Y || !q || !u || M <= 0 || ldy < M || I <= 0 || (u_old && !du2)) { set_error("rowdot: bad argument"); return CMTFPLS_EINVAL; }
  if (u_old && (!ws || ws_bytes < kSmallBlocks * sizeof(double))) { set_error("rowdot: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = u_old ? static_cast<double*>(ws) : nullptr;
  hipLaunchKernelGGL(rowdot_kernel, dim3(kSmallBlocks), dim3(256), 0, st, Y, ldy, M, I, q, u, u_old, part);
  if (u_old) hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, part, (int64_t)kSmallBlocks, du2);
  return check_launch("rowdot");
}

int cmtfpls_y_deflate_f64(double* Y, int ldy, int M, int64_t I, const double* T, int ldt, int R,
                          const double* b, const double* q, double* ssq, void* ws, size_t ws_bytes, void* stream) {
  if (!Y || !T || !b || !q || !ssq || M <= 0 || ldy < M || R <= 0 || ldt < R || I <= 0) { set_error("y_deflate: bad argument"); return CMTFPLS_EINVAL; }
  if (!ws || ws_bytes < kSmallBlocks * sizeof(double)) { set_error("y_deflate: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(y_deflate_kernel, dim3(kSmallBlocks), dim3(256), 0, st, Y, ldy, M, I, T, ldt, R, b, q, part);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, part, (int64_t)kSmallBlocks, ssq);
  return check_launch("y_deflate");
}

int cmtfpls_sum_f64(const double* in, int64_t n, double* out, void* stream) {
  if (!in || !out || n <= 0) { set_error("sum: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, n, out);
  return check_launch("sum");
}

int cmtfpls_normalize_f64(double* v, int64_t n, double* nrm, void* stream) {
  if (!v || n <= 0) { set_error("normalize: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, nrm);
  return check_launch("normalize");
}

int cmtfpls_q_update_f64(const double* qpart, int nblk, int M, double* q, int normalize, const double* G, const double* q_prev,
                         double* du2, void* stream) {
  if (!q || M <= 0 || (qpart && nblk <= 0) || (G && (!q_prev || !du2))) { set_error("q_update: bad argument"); return CMTFPLS_EINVAL; }
  if (M > 64) { set_error("q_update: more than 64 responses; use gram_tn + normalize + rowdot"); return CMTFPLS_EUNSUPPORTED; }
  hipLaunchKernelGGL(q_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, qpart, nblk, M, q, normalize, G, q_prev, du2);
  return check_launch("q_update");
}

int cmtfpls_s_downdate_f64(double* S, int M, int A, int B, const double* ya, const double* wA, const double* wB,
                           const double* q, const double* v, void* stream) {
  if (!S || !ya || !wA || !wB || !q || !v || M <= 0 || A <= 0 || B <= 0) { set_error("s_downdate: bad argument"); return CMTFPLS_EINVAL; }
  const int64_t P = (int64_t)A * B;
  hipLaunchKernelGGL(s_downdate_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, M, P, B, ya, wA, wB, q, v);
  return check_launch("s_downdate");
}

int cmtfpls_axpy_scalar_f64(double* y, int64_t n, const double* a, const double* x, void* stream) {
  if (!y || !a || n <= 0) { set_error("axpy_scalar: bad argument"); return CMTFPLS_EINVAL; }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(axpy_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, n, a, x);
  return check_launch("axpy_scalar");
}

int cmtfpls_kr_axpy_f64(double* v, int A, int B, const double* WA, const double* WB, int ld, int k, const double* coef, void* stream) {
  if (!v || !WA || !WB || !coef || A <= 0 || B <= 0 || k < 0 || ld < k) { set_error("kr_axpy: bad argument"); return CMTFPLS_EINVAL; }
  if (k > 64) { set_error("kr_axpy: more than 64 terms"); return CMTFPLS_EUNSUPPORTED; }
  if (k == 0) return CMTFPLS_OK;
  const int64_t P = (int64_t)A * B;
  hipLaunchKernelGGL(kr_axpy_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, P, B, WA, WB, ld, k, coef);
  return check_launch("kr_axpy");
}

// One inner iteration of the cross-covariance form issued by ONE host call (its kernels are tiny: a
// Python-side launch per kernel leaves the GPU idle between them).  Calls the entries above in order.
int cmtfpls_xcov_iterate_f64(const double* S, int M, int A, int B, const double* q_cur, double* Z, double* wA, double* wB,
                             double* info, int n_squarings, double* q_new, const double* G, double* du2, int first,
                             void* ws_contract, size_t ws_contract_bytes, void* ws_rank1, size_t ws_rank1_bytes, void* stream) {
  if (!S || !q_cur || !Z || !wA || !wB || !q_new || !G || !du2 || M <= 0 || A <= 0 || B <= 0) {
    set_error("xcov_iterate: bad argument");
    return CMTFPLS_EINVAL;
  }
  if (M > 64) { set_error("xcov_iterate: more than 64 responses"); return CMTFPLS_EUNSUPPORTED; }
  const int64_t P = (int64_t)A * B;
  int rc = CMTFPLS_OK;
  if (first) {                                                                                                // Z = sum_m q_m S_m
    hipLaunchKernelGGL(s_contract_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, M, P, q_cur, Z);
    rc = check_launch("xcov_iterate: s_contract");
  }
  // rank-1 of Z, then Y^T t = S (wA (x) wB): the extraction's last kernel and the score share a launch
  if (rc == CMTFPLS_OK) rc = cmtfpls_rank1_score_f64(Z, A, B, wA, wB, info, n_squarings, S, M, q_new, ws_rank1, ws_rank1_bytes, stream);
  if (rc == CMTFPLS_OK) rc = cmtfpls_q_update_f64(nullptr, 0, M, q_new, 1, G, q_cur, du2, stream);            // / norm, |du|^2
  return rc;
}

int cmtfpls_xcov_iterate_blocks_f64(const cmtfpls_xcov_block* blocks, int nb, int M, const double* q_cur, double* tq, double* q_new,
                                    const double* G, double* du2, int first, void* ws_rank1, size_t ws_rank1_bytes, void* stream) {
  if (!blocks || nb <= 0 || !q_cur || !tq || !q_new || !G || !du2 || M <= 0) { set_error("xcov_iterate_blocks: bad argument"); return CMTFPLS_EINVAL; }
  if (M > 64) { set_error("xcov_iterate_blocks: more than 64 responses"); return CMTFPLS_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  int rc = CMTFPLS_OK;
  for (int b = 0; b < nb && rc == CMTFPLS_OK; ++b) {
    const cmtfpls_xcov_block& k = blocks[b];
    if (!k.S || !k.Z || !k.wA || !k.wB || k.A <= 0 || k.B <= 0 || (k.order != 2 && k.order != 3) || (k.order == 2 && k.A != 1) ||
        (k.order == 3 && !k.info)) {
      set_error("xcov_iterate_blocks: bad block");
      return CMTFPLS_EINVAL;
    }
    const int64_t P = (int64_t)k.A * k.B;
    if (k.order == 2 && P <= 8192) {                       // a matrix block: its whole share of the iteration in one launch
      hipLaunchKernelGGL(s_vector_block_kernel, dim3(1), dim3(1024), (size_t)P * sizeof(double), st, k.S, k.S2 ? k.S2 : k.S, k.colcnt,
                         k.n_samples, M, (int)P, q_cur, k.Z, k.wB, tq + (int64_t)b * M);
      rc = check_launch("xcov_iterate_blocks: s_vector_block");
      continue;
    }
    if (first) {
      hipLaunchKernelGGL(s_contract_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, k.S, M, P, q_cur, k.Z);   // cmtf.py:94
      rc = check_launch("xcov_iterate_blocks: s_contract");
      if (rc == CMTFPLS_OK && k.colcnt) rc = cmtfpls_colscale_f64(k.Z, P, k.colcnt, k.n_samples, stream);             // missingvals.py:17-19
    }
    if (rc != CMTFPLS_OK) break;
    if (k.order == 3) {                                    // cmtf.py:98-104, then cmtf.py:106-119 on S2 (same launch when the shape allows)
      rc = cmtfpls_rank1_score_f64(k.Z, k.A, k.B, k.wA, k.wB, k.info, k.n_squarings, k.S2 ? k.S2 : k.S, M, tq + (int64_t)b * M, ws_rank1,
                                   ws_rank1_bytes, stream);
      continue;
    } else {
      hipLaunchKernelGGL(normalize_to_kernel, dim3(1), dim3(1024), 0, st, k.Z, k.wB, P);
      rc = check_launch("xcov_iterate_blocks: normalize_to");
    }
    if (rc == CMTFPLS_OK) rc = cmtfpls_score_s_f64(k.S2 ? k.S2 : k.S, M, k.A, k.B, k.wA, k.wB, tq + (int64_t)b * M, stream);   // cmtf.py:106-119
  }
  // cmtf.py:120-125: q_new = mean_b tq_b, normalised -- the SUM of the rows normalised is the same vector (the 1 / nb drops out of
  // q / |q|), and summing the rows of a small matrix is what q_update does with its partials: one launch for both steps
  if (rc == CMTFPLS_OK) rc = cmtfpls_q_update_f64(tq, nb, M, q_new, 1, G, q_cur, du2, stream);
  return rc;
}

int cmtfpls_colscale_f64(double* Z, int64_t P, const double* colcnt, double n_samples, void* stream) {
  if (!Z || !colcnt || P <= 0) { set_error("colscale: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(colscale_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, P, colcnt, n_samples);
  return check_launch("colscale");
}

int cmtfpls_scores_mean_f64(const double* Ts, int nb, int64_t I, double* out, void* stream) {
  if (!Ts || !out || nb <= 0 || I <= 0) { set_error("scores_mean: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(scores_mean_kernel, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Ts, nb, I, out);
  return check_launch("scores_mean");
}

}  // extern "C"
