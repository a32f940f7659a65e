// Rank-1 CP factors of a cross-covariance TENSOR Z of order 3 to 7 (X of order 4 to 8; round 2: order 3 or 4):
//   parafac(Z, 1, tol=tol, init="svd", normalize_factors=True)[1]      tpls.py:86-88, cmtf.py:100-102
// restated from tensorly 0.9.0's published algorithm exactly as oracle/nipals_oracle.py does
// (value-level parity with tensorly itself is UNPINNED):
//   init    f_m = leading left singular vector of the mode-m unfolding, largest-|.| entry positive;
//   sweep   for every mode m: f_m = (Z x_{i != m} f_i) * weight / (weight^2 prod_{i != m} f_i.f_i);
//           rec_error = sqrt(|Z|^2 + |weight * f_0 o f_1 o ...|^2 - 2 <Z, .>) / |Z|;
//           stop when |rec_error - previous| < tol from the 2nd sweep on, at most 100 sweeps;
//           otherwise normalise every f_m and fold the norms into weight.
// These tensors are tiny next to X (prod of the trailing dims), so the ALS runs in ONE workgroup
// with the factors in LDS and Z read from L2; the init reuses the matrix rank-1 kernels on each
// materialised unfolding.
#include "common.hpp"

namespace cmtfpls {

constexpr int kMaxOrder = 7;

struct TensorDims {
  int n;                 // order of Z (3 .. kMaxOrder)
  int d[kMaxOrder];      // dims
  int64_t stride[kMaxOrder];
  int64_t total;
};

// out (d[mode] x total/d[mode], row-major) = mode-`mode` unfolding of Z (C order of the remaining modes)
__global__ __launch_bounds__(256) void unfold_kernel(const double* __restrict__ Z, TensorDims td, int mode, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= td.total) return;
  // e enumerates the OUTPUT: row a = index along `mode`, column = C-order index over the other modes
  const int64_t cols = td.total / td.d[mode];
  const int a = (int)(e / cols);
  int64_t c = e % cols;
  int64_t off = (int64_t)a * td.stride[mode];
  for (int i = td.n - 1; i >= 0; --i) {
    if (i == mode) continue;
    off += (c % td.d[i]) * td.stride[i];
    c /= td.d[i];
  }
  out[e] = Z[off];
}

// out[a] = sum over all other indices of Z[...] * prod_{i != mode} f_i[idx_i]      (one wavefront per a)
__device__ void mode_contract(const double* __restrict__ Z, const TensorDims& td, int mode, double* const* f, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t cols = td.total / td.d[mode];
  for (int a = wv; a < td.d[mode]; a += nw) {
    double s = 0.0;
    for (int64_t c0 = lane; c0 < cols; c0 += 64) {
      int64_t c = c0, off = (int64_t)a * td.stride[mode];
      double w = 1.0;
      for (int i = td.n - 1; i >= 0; --i) {
        if (i == mode) continue;
        const int idx = (int)(c % td.d[i]);
        c /= td.d[i];
        off += idx * td.stride[i];
        w *= f[i][idx];
      }
      s = fma(Z[off], w, s);
    }
    s = wave_sum(s);
    if (lane == 0) out[a] = s;
  }
  __syncthreads();
}

__device__ double vec_dot(const double* a, const double* b, int n, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s = fma(a[i], b[i], s);
  s = block_sum(s, red);
  __syncthreads();
  return s;
}

// init: (N, max d) factor matrix `f0` in global (from the per-mode matrix rank-1 calls); out: f_out
__global__ __launch_bounds__(1024) void cp_rank1_als_kernel(const double* __restrict__ Z, TensorDims td,
                                                           const double* __restrict__ f_init, int ld_init, double tol,
                                                           int max_sweeps, double* __restrict__ f_out, int ld_out,
                                                           double* __restrict__ info) {
  extern __shared__ double lds[];
  __shared__ double red[16];
  __shared__ int s_arg;
  double* f[kMaxOrder];
  double* tmp;
  {
    double* p = lds;
    for (int m = 0; m < kMaxOrder; ++m) { f[m] = p; if (m < td.n) p += (td.d[m] + 1) & ~1; }
    tmp = p;   // max d doubles
  }
  // load the init vectors, apply the sign rule (largest-|.| entry positive, first on ties)
  for (int m = 0; m < td.n; ++m) {
    for (int i = threadIdx.x; i < td.d[m]; i += blockDim.x) f[m][i] = f_init[(int64_t)m * ld_init + i];
    __syncthreads();
    if (threadIdx.x == 0) {
      int arg = 0;
      double best = fabs(f[m][0]);
      for (int i = 1; i < td.d[m]; ++i) { const double v = fabs(f[m][i]); if (v > best) { best = v; arg = i; } }
      s_arg = (f[m][arg] < 0.0) ? 1 : 0;
    }
    __syncthreads();
    if (s_arg) for (int i = threadIdx.x; i < td.d[m]; i += blockDim.x) f[m][i] = -f[m][i];
    __syncthreads();
  }
  // |Z|^2
  double zz = 0.0;
  for (int64_t e = threadIdx.x; e < td.total; e += blockDim.x) zz = fma(Z[e], Z[e], zz);
  zz = block_sum(zz, red);
  __syncthreads();
  const double norm_z = sqrt(zz);

  double weight = 1.0, prev_err = 0.0;
  int sweeps = 0;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    ++sweeps;
    double iprod = 0.0;
    for (int m = 0; m < td.n; ++m) {
      double gram = weight * weight;
      for (int i = 0; i < td.n; ++i)
        if (i != m) gram *= vec_dot(f[i], f[i], td.d[i], red);
      mode_contract(Z, td, m, f, tmp);                       // tmp = Z x_{i != m} f_i   (without weight)
      if (m == td.n - 1) {
        // <mttkrp, new f_last> * weight with mttkrp = tmp * weight and f_last = mttkrp / gram
        const double tt = vec_dot(tmp, tmp, td.d[m], red);
        iprod = (weight * weight * tt / gram) * weight;
      }
      for (int i = threadIdx.x; i < td.d[m]; i += blockDim.x) f[m][i] = tmp[i] * weight / gram;
      __syncthreads();
    }
    double fn2 = weight * weight;
    for (int i = 0; i < td.n; ++i) fn2 *= vec_dot(f[i], f[i], td.d[i], red);
    const double err = sqrt(fabs(norm_z * norm_z + fn2 - 2.0 * iprod)) / norm_z;
    if (sweep >= 1 && fabs(prev_err - err) < tol) break;
    prev_err = err;
    for (int i = 0; i < td.n; ++i) {
      const double nrm = sqrt(vec_dot(f[i], f[i], td.d[i], red));
      weight *= nrm;
      for (int j = threadIdx.x; j < td.d[i]; j += blockDim.x) f[i][j] = f[i][j] / nrm;
      __syncthreads();
    }
  }
  for (int m = 0; m < td.n; ++m)
    for (int i = threadIdx.x; i < td.d[m]; i += blockDim.x) f_out[(int64_t)m * ld_out + i] = f[m][i];
  if (info && threadIdx.x == 0) { info[0] = 1.0; info[1] = (double)sweeps; }
}

// out[c] = a[c / nb] * b[c % nb]    (Kronecker product of two vectors, C order)
__global__ __launch_bounds__(256) void kron_kernel(const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
                                                  double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < (int64_t)na * nb) out[c] = a[c / nb] * b[c % nb];
}

static bool make_dims(const int* dims, int n, TensorDims* td) {
  if (n < 3 || n > kMaxOrder) return false;
  td->n = n;
  int64_t s = 1;
  for (int i = n - 1; i >= 0; --i) {
    if (dims[i] <= 0) return false;
    td->d[i] = dims[i];
    td->stride[i] = s;
    s *= dims[i];
  }
  for (int i = n; i < kMaxOrder; ++i) { td->d[i] = 1; td->stride[i] = 0; }
  td->total = s;
  return true;
}

static inline size_t align_up_t(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_rank1_workspace_bytes(int A, int B);
int cmtfpls_rank1_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                      int n_squarings, void* ws, size_t ws_bytes, void* stream);

size_t cmtfpls_rank1_tensor_workspace_bytes(const int* dims, int n) {
  TensorDims td;
  if (!dims || !make_dims(dims, n, &td)) return 0;
  size_t ws_mat = 0, maxd = 0;
  for (int m = 0; m < n; ++m) {
    const size_t w = cmtfpls_rank1_workspace_bytes(td.d[m], (int)(td.total / td.d[m]));
    if (w > ws_mat) ws_mat = w;
    if ((size_t)td.d[m] > maxd) maxd = td.d[m];
  }
  return align_up_t(ws_mat, 256) + align_up_t((size_t)td.total * sizeof(double), 256) /* unfolding */ +
         align_up_t((size_t)td.total * sizeof(double), 256) /* discarded right vector */ + align_up_t((size_t)n * maxd * sizeof(double), 256);
}

int cmtfpls_rank1_tensor_f64(const double* Z, const int* dims, int n, double tol, double* factors, int ld,
                             double* info, int n_squarings, void* ws, size_t ws_bytes, void* stream) {
  TensorDims td;
  if (!Z || !dims || !factors || !make_dims(dims, n, &td)) { set_error("rank1_tensor: bad argument (order 3 to 7)"); return CMTFPLS_EINVAL; }
  size_t maxd = 0, sumd = 0;
  for (int m = 0; m < n; ++m) {
    if ((size_t)td.d[m] > maxd) maxd = td.d[m];
    sumd += (td.d[m] + 1) & ~1;
    if (td.d[m] > 1024 || td.total / td.d[m] > (1 << 30)) { set_error("rank1_tensor: mode too large"); return CMTFPLS_EUNSUPPORTED; }
  }
  if ((size_t)ld < maxd) { set_error("rank1_tensor: ld < max dim"); return CMTFPLS_EINVAL; }
  if (!ws || ws_bytes < cmtfpls_rank1_tensor_workspace_bytes(dims, n)) { set_error("rank1_tensor: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  size_t ws_mat = 0;
  for (int m = 0; m < n; ++m) {
    const size_t w = cmtfpls_rank1_workspace_bytes(td.d[m], (int)(td.total / td.d[m]));
    if (w > ws_mat) ws_mat = w;
  }
  char* p = static_cast<char*>(ws);
  void* wsm = p;
  p += align_up_t(ws_mat, 256);
  double* unf = reinterpret_cast<double*>(p);
  p += align_up_t((size_t)td.total * sizeof(double), 256);
  double* vright = reinterpret_cast<double*>(p);
  p += align_up_t((size_t)td.total * sizeof(double), 256);
  double* finit = reinterpret_cast<double*>(p);
  // init: leading left singular vector of every unfolding (tensorly initialize_cp, init="svd")
  for (int m = 0; m < n; ++m) {
    const int rows = td.d[m], cols = (int)(td.total / td.d[m]);
    const double* M = Z;
    if (m != 0) {
      hipLaunchKernelGGL(unfold_kernel, dim3((unsigned)((td.total + 255) / 256)), dim3(256), 0, st, Z, td, m, unf);
      M = unf;
    }
    const int rc = cmtfpls_rank1_f64(M, rows, cols, finit + (size_t)m * maxd, vright, nullptr, nullptr, n_squarings, wsm, ws_mat, stream);
    if (rc != CMTFPLS_OK) return rc;
  }
  const size_t lds = (sumd + ((maxd + 1) & ~(size_t)1)) * sizeof(double);
  hipLaunchKernelGGL(cp_rank1_als_kernel, dim3(1), dim3(1024), lds, st, Z, td, finit, (int)maxd, tol, 100, factors, ld, info);
  return check_launch("rank1_tensor");
}

int cmtfpls_kron_f64(const double* a, int na, const double* b, int nb, double* out, void* stream) {
  if (!a || !b || !out || na <= 0 || nb <= 0) { set_error("kron: bad argument"); return CMTFPLS_EINVAL; }
  const int64_t tot = (int64_t)na * nb;
  hipLaunchKernelGGL(kron_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
  return check_launch("kron");
}

}  // extern "C"
