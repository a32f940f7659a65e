// Tiny dense f64 algebra that used to round-trip through the host between two X sweeps:
//   normal_solve      : coef_[:, a] = lstsq(T, u)  (tpls.py:110-112) from the (a+1) x (a+1) normal equations
//   unit_upper_solve  : T (I + triu(G, 1)) = M     (the R x R fix-up of the one-pass projection, engine.project)
//   kr_gram           : G (.)= L^T L               (Gram of a Khatri-Rao product = Hadamard product of the mode Grams)
//   khatri_rao        : column-wise Kronecker product of two loading matrices
// Each is one workgroup (or one thread per row) of plain f64 code: the sizes are R <= 64, launches are
// latency, and the point is that the component epilogue has no device -> host -> device hop.
#include "common.hpp"

namespace cmtfpls {

constexpr int kSolveMax = 64;

// Solve G b = g for symmetric positive semi-definite G (k x k, row-major, k <= 64) by a Cholesky
// factorisation of the EQUILIBRATED matrix D G D, D = diag(G)^(-1/2): the scores' columns differ in scale by
// orders of magnitude (late components of a well-explained X), which the reference's lstsq on T itself
// tolerates (rcond = machine precision relative to T's largest singular value); solving the raw normal
// equations would square that spread.  A column whose pivot falls below k * eps after equilibration is linearly
// dependent on the earlier ones to working precision (or identically zero): its coefficient is set to 0 and it
// is dropped from the system, which is what a truncated least-squares solve does with it.
__global__ __launch_bounds__(kSolveMax) void normal_solve_kernel(const double* __restrict__ G, const double* __restrict__ g, int k,
                                                                double* __restrict__ b, int incb) {
  __shared__ double A[kSolveMax][kSolveMax + 1];
  __shared__ double d[kSolveMax], y[kSolveMax];
  __shared__ int dep[kSolveMax];
  const int i = threadIdx.x;
  if (i < k) {
    const double gii = G[(int64_t)i * k + i];
    d[i] = (gii > 0.0 && isfinite(gii)) ? 1.0 / sqrt(gii) : 0.0;
    dep[i] = 0;
  }
  __syncthreads();
  if (i < k) {
    for (int j = 0; j < k; ++j) A[i][j] = G[(int64_t)i * k + j] * d[i] * d[j];
    y[i] = g[i] * d[i];
  }
  __syncthreads();
  const double tiny = (double)k * 2.220446049250313e-16;
  for (int c = 0; c < k; ++c) {
    const double piv = A[c][c];
    const bool ok = piv > tiny;               // uniform: every thread reads the same LDS word
    __syncthreads();
    if (ok) {
      const double l = sqrt(piv);
      if (i == c) A[c][c] = l;
      if (i > c && i < k) A[i][c] = A[i][c] / l;
    } else {
      if (i == c) { A[c][c] = 1.0; dep[c] = 1; }
      if (i > c && i < k) A[i][c] = 0.0;
    }
    __syncthreads();
    if (ok && i > c && i < k) {
      const double lic = A[i][c];
      for (int j = c + 1; j <= i; ++j) A[i][j] -= lic * A[j][c];
    }
    __syncthreads();
  }
  // forward L z = y, backward L^T x = z (thread 0: k <= 64, a few hundred dependent flops)
  if (i == 0) {
    for (int r = 0; r < k; ++r) {
      double s = y[r];
      for (int j = 0; j < r; ++j) s -= A[r][j] * y[j];
      y[r] = dep[r] ? 0.0 : s / A[r][r];
    }
    for (int r = k - 1; r >= 0; --r) {
      double s = y[r];
      for (int j = r + 1; j < k; ++j) s -= A[j][r] * y[j];
      y[r] = dep[r] ? 0.0 : s / A[r][r];
    }
  }
  __syncthreads();
  if (i < k) b[(int64_t)i * incb] = y[i] * d[i];
}

// The same solve for 64 < k <= kSolveBigMax (round 3; the reference's lstsq has no limit on n_components, tpls.py:110-112):
// identical algorithm and pivot rule, the equilibrated matrix in a k x (k + 1) global-memory workspace instead of the
// LDS, one 256-thread workgroup, thread t owning rows t, t + 256, ...; the two triangular solves are column sweeps with
// one barrier per column.  Only fits with more than 64 components come here: a few hundred microseconds per component.
constexpr int kSolveBigMax = 1024;
__global__ __launch_bounds__(256) void normal_solve_big_kernel(const double* __restrict__ G, const double* __restrict__ g, int k,
                                                               double* __restrict__ b, int incb, double* __restrict__ Aw) {
  const int ld = k + 1;
  double* d = Aw + (size_t)k * ld;
  double* y = d + k;
  double* depf = y + k;                       // 1.0 = dependent column
  __shared__ double s_piv;
  const int t = threadIdx.x;
  for (int i = t; i < k; i += 256) {
    const double gii = G[(int64_t)i * k + i];
    d[i] = (gii > 0.0 && isfinite(gii)) ? 1.0 / sqrt(gii) : 0.0;
    depf[i] = 0.0;
  }
  __syncthreads();
  for (int i = t; i < k; i += 256) {
    for (int j = 0; j < k; ++j) Aw[(size_t)i * ld + j] = G[(int64_t)i * k + j] * d[i] * d[j];
    y[i] = g[i] * d[i];
  }
  __syncthreads();
  const double tiny = (double)k * 2.220446049250313e-16;
  for (int c = 0; c < k; ++c) {
    if (t == 0) s_piv = Aw[(size_t)c * ld + c];
    __syncthreads();
    const double piv = s_piv;
    const bool ok = piv > tiny;
    const double l = ok ? sqrt(piv) : 1.0;
    for (int i = c + t; i < k; i += 256) {
      if (i == c) { Aw[(size_t)c * ld + c] = l; if (!ok) depf[c] = 1.0; }
      else Aw[(size_t)i * ld + c] = ok ? Aw[(size_t)i * ld + c] / l : 0.0;
    }
    __syncthreads();
    if (ok)
      for (int i = c + 1 + t; i < k; i += 256) {
        const double lic = Aw[(size_t)i * ld + c];
        for (int j = c + 1; j <= i; ++j) Aw[(size_t)i * ld + j] -= lic * Aw[(size_t)j * ld + c];
      }
    __syncthreads();
  }
  // forward L z = y (column sweep), backward L^T x = z (row r of L^T is column r of L)
  for (int r = 0; r < k; ++r) {
    if (t == 0) y[r] = (depf[r] != 0.0) ? 0.0 : y[r] / Aw[(size_t)r * ld + r];
    __syncthreads();
    const double yr = y[r];
    for (int j = r + 1 + t; j < k; j += 256) y[j] -= Aw[(size_t)j * ld + r] * yr;
    __syncthreads();
  }
  for (int r = k - 1; r >= 0; --r) {
    if (t == 0) y[r] = (depf[r] != 0.0) ? 0.0 : y[r] / Aw[(size_t)r * ld + r];
    __syncthreads();
    const double yr = y[r];
    for (int j = t; j < r; j += 256) y[j] -= Aw[(size_t)r * ld + j] * yr;
    __syncthreads();
  }
  for (int i = t; i < k; i += 256) b[(int64_t)i * incb] = y[i] * d[i];
}

// rows of T solve  T (I + triu(U, 1)) = M - 1 shift^T:  t_a = (m_a - shift_a) - sum_{j < a} t_j U[j][a]   (one thread per
// row, in place).  shift (R, nullable): the centring of an UNCENTRED MTTKRP, (X - 1 mean^T) W = X W - 1 (mean^T W)^T, so
// that transform / predict read the caller's X once and never write a centred copy.  nan_flag (nullable, zeroed by the
// caller): set to 1 when any entry of M is NaN -- a missing value somewhere in that row of X (NaN survives every product
// and sum), which tells the caller to take the masked sequential path instead.
__global__ __launch_bounds__(256) void unit_upper_solve_rows_kernel(double* __restrict__ Mx, int64_t I, int ld, int R,
                                                                   const double* __restrict__ U, const double* __restrict__ shift,
                                                                   int* __restrict__ nan_flag) {
  __shared__ double Us[kSolveMax * kSolveMax];
  __shared__ double sh[kSolveMax];
  for (int idx = threadIdx.x; idx < R * R; idx += 256) Us[idx] = U[idx];
  for (int idx = threadIdx.x; idx < R; idx += 256) sh[idx] = shift ? shift[idx] : 0.0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= I) return;
  double* row = Mx + i * ld;                 // t_j (j < a) is read back from the row this thread just wrote
  bool bad = false;
#pragma unroll 1
  for (int a = 0; a < R; ++a) {
    double s = row[a];
    bad |= (s != s);
    s -= sh[a];
    for (int j = 0; j < a; ++j) s -= row[j] * Us[j * R + a];
    row[a] = s;
  }
  if (bad && nan_flag) *nan_flag = 1;        // every writer stores the same value
}

// G[r][s] = (first ? 1 : G[r][s]) * scale * sum_j L[j][r] L[j][s]     (L: n x R row-major; one workgroup)
__global__ __launch_bounds__(256) void kr_gram_kernel(const double* __restrict__ L, int n, int R, double* __restrict__ G,
                                                     int first, double scale) {
  for (int o = threadIdx.x; o < R * R; o += 256) {
    const int r = o / R, s = o % R;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(L[(int64_t)j * R + r], L[(int64_t)j * R + s], acc);
    G[o] = (first ? 1.0 : G[o]) * scale * acc;
  }
}

// g[j] = (first ? 1 : g[j]) * sum_i L[i][j] L[i][a],  j < a: ROW a of the Gram matrix above, all the never-writing cross-covariance
// loop needs per component (w_j^T w_a, j < a) -- a launches' worth of latency instead of the R x R matrix by one lane per entry
// walking n rows (22 us at 128 x 10).  One workgroup; 16 rows of loads in flight per lane.
__global__ __launch_bounds__(256) void kr_gram_row_kernel(const double* __restrict__ L, int n, int R, int a, double* __restrict__ g, int first) {
  for (int j = threadIdx.x; j < a; j += 256) {
    double acc = 0.0;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
      double x[16], y[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        x[u] = L[(int64_t)(i + u) * R + j];
        y[u] = L[(int64_t)(i + u) * R + a];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fma(x[u], y[u], acc);
    }
    for (; i < n; ++i) acc = fma(L[(int64_t)i * R + j], L[(int64_t)i * R + a], acc);
    g[j] = (first ? 1.0 : g[j]) * acc;
  }
}

// out[(j * nb + k) * R + r] = Am[j * R + r] * Bm[k * R + r]
__global__ __launch_bounds__(256) void khatri_rao_kernel(const double* __restrict__ Am, int na, const double* __restrict__ Bm, int nb,
                                                        int R, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = (int64_t)na * nb * R;
  if (idx >= tot) return;
  const int r = (int)(idx % R);
  const int64_t jk = idx / R;
  out[idx] = Am[(jk / nb) * R + r] * Bm[(jk % nb) * R + r];
}

// out[i, m] = mean[m] + sum_a S[i, a] Bm[a, m]: the last line of predict, `X_projection @ coef_ @ Q^T + Y_mean`
// (tpls.py:143, cmtf.py:177), with Bm = coef_ Q^T (R x M) formed by the caller; one thread per output element.
__global__ __launch_bounds__(256) void predict_rows_kernel(const double* __restrict__ S, int lds_, int R, const double* __restrict__ Bm,
                                                          const double* __restrict__ mean, double* __restrict__ out, int ldo, int64_t I, int M) {
  extern __shared__ double sb[];                    // Bm (R x M) then mean (M)
  for (int idx = threadIdx.x; idx < R * M; idx += 256) sb[idx] = Bm[idx];
  for (int idx = threadIdx.x; idx < M; idx += 256) sb[R * M + idx] = mean ? mean[idx] : 0.0;
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= I * M) return;
  const int64_t i = o / M;
  const int m = (int)(o % M);
  double acc = 0.0;
  for (int a = 0; a < R; ++a) acc = fma(S[i * lds_ + a], sb[a * M + m], acc);
  out[i * ldo + m] = acc + sb[R * M + m];
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

int cmtfpls_normal_solve_f64(const double* G, const double* g, int k, double* b, int incb, void* stream) {
  if (!G || !g || !b || k <= 0 || incb <= 0) { set_error("normal_solve: bad argument"); return CMTFPLS_EINVAL; }
  if (k > kSolveMax) { set_error("normal_solve: more than 64 components; use cmtfpls_normal_solve_ws_f64"); return CMTFPLS_EUNSUPPORTED; }
  hipLaunchKernelGGL(normal_solve_kernel, dim3(1), dim3(kSolveMax), 0, (hipStream_t)stream, G, g, k, b, incb);
  return check_launch("normal_solve");
}

size_t cmtfpls_normal_solve_workspace_bytes(int k) {
  if (k <= kSolveMax || k > kSolveBigMax) return 0;
  return ((size_t)k * (k + 1) + 3 * (size_t)k) * sizeof(double);
}

int cmtfpls_normal_solve_ws_f64(const double* G, const double* g, int k, double* b, int incb, void* ws, size_t ws_bytes, void* stream) {
  if (k <= kSolveMax) return cmtfpls_normal_solve_f64(G, g, k, b, incb, stream);
  if (!G || !g || !b || incb <= 0) { set_error("normal_solve: bad argument"); return CMTFPLS_EINVAL; }
  if (k > kSolveBigMax) { set_error("normal_solve: more than 1024 components"); return CMTFPLS_EUNSUPPORTED; }
  if (!ws || ws_bytes < cmtfpls_normal_solve_workspace_bytes(k)) { set_error("normal_solve: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipLaunchKernelGGL(normal_solve_big_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, G, g, k, b, incb, static_cast<double*>(ws));
  return check_launch("normal_solve");
}

int cmtfpls_unit_upper_solve_rows_f64(double* Mx, int64_t I, int ld, int R, const double* U, const double* shift, int* nan_flag,
                                      void* stream) {
  if (!Mx || !U || I <= 0 || R <= 0 || ld < R) { set_error("unit_upper_solve_rows: bad argument"); return CMTFPLS_EINVAL; }
  if (R > kSolveMax) { set_error("unit_upper_solve_rows: more than 64 components"); return CMTFPLS_EUNSUPPORTED; }
  hipLaunchKernelGGL(unit_upper_solve_rows_kernel, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Mx, I, ld, R, U,
                     shift, nan_flag);
  return check_launch("unit_upper_solve_rows");
}

int cmtfpls_kr_gram_f64(const double* L, int n, int R, double* G, int first, double scale, void* stream) {
  if (!L || !G || n <= 0 || R <= 0) { set_error("kr_gram: bad argument"); return CMTFPLS_EINVAL; }
  hipLaunchKernelGGL(kr_gram_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, L, n, R, G, first, scale);
  return check_launch("kr_gram");
}

int cmtfpls_kr_gram_row_f64(const double* L, int n, int R, int a, double* g, int first, void* stream) {
  if (!L || !g || n <= 0 || R <= 0 || a < 0 || a >= R) { set_error("kr_gram_row: bad argument"); return CMTFPLS_EINVAL; }
  if (a == 0) return CMTFPLS_OK;
  hipLaunchKernelGGL(kr_gram_row_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, L, n, R, a, g, first);
  return check_launch("kr_gram_row");
}

int cmtfpls_predict_rows_f64(const double* S, int64_t I, int lds_, int R, const double* Bm, int M, const double* mean, double* out, int ldo,
                             void* stream) {
  if (!S || !Bm || !out || I <= 0 || R <= 0 || M <= 0 || lds_ < R || ldo < M) { set_error("predict_rows: bad argument"); return CMTFPLS_EINVAL; }
  const size_t lds_bytes = ((size_t)R * M + M) * sizeof(double);
  if (lds_bytes > 64 * 1024) { set_error("predict_rows: coef_ Q^T does not fit the LDS"); return CMTFPLS_EUNSUPPORTED; }
  const int64_t tot = I * M;
  hipLaunchKernelGGL(predict_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), lds_bytes, (hipStream_t)stream, S, lds_, R, Bm, mean,
                     out, ldo, I, M);
  return check_launch("predict_rows");
}

int cmtfpls_khatri_rao_f64(const double* Am, int na, const double* Bm, int nb, int R, double* out, void* stream) {
  if (!Am || !Bm || !out || na <= 0 || nb <= 0 || R <= 0) { set_error("khatri_rao: bad argument"); return CMTFPLS_EINVAL; }
  const int64_t tot = (int64_t)na * nb * R;
  hipLaunchKernelGGL(khatri_rao_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Am, na, Bm, nb, R, out);
  return check_launch("khatri_rao");
}

}  // extern "C"
