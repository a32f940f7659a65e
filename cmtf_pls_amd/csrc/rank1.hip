// Rank-1 extraction of the cross-covariance matrix Z (A x B, f64): the leading singular pair,
// i.e. what parafac(Z, 1, init="svd", normalize_factors=True) returns for a matrix
// (tpls.py:86-88, cmtf.py:100-102).  LAPACK is not available on the device and a plain power
// iteration converges at (s2/s1)^2 per step, so the Gram matrix G of the smaller side is squared
// repeatedly instead: after s squarings G^(2^s) is rank one to (s2/s1)^(2^(s+1)), each squaring is
// one n x n x n f64 product (n <= 256 for the benchmark shapes) spread over n^2/256 workgroups.
// Scaling between squarings is by an exact power of two taken from the trace, so the iteration is
// bit-reproducible; it stops early (later launches return at once) when tr(G^2) == tr(G)^2 to
// 1e-13.  The dominant column of the final G then seeds two exact power steps with Z itself, which
// also gives exact zeros in the loadings wherever Z has an all-zero row or column
// (tests/test_tpls.py:98-104 relies on that).
#include "common.hpp"

namespace cmtfpls {

constexpr int kTile = 16;
constexpr int kMaxTiles = 64;   // n <= 1024
constexpr int kMaxSteps = 48;

struct Rank1Ctl {
  double trace[kMaxSteps + 2][kMaxTiles];  // trace[s][tile]: diagonal-tile partial traces of G_s
  int done;                                // set once G is numerically rank one
  int final_buf;                           // which ping-pong buffer holds the final G
};

__device__ __forceinline__ double pow2_scale_from_trace(const double* parts, int nt, double* tr_out) {
  double tr = 0.0;
  for (int i = 0; i < nt; ++i) tr += parts[i];
  *tr_out = tr;
  if (!(tr > 0.0) || !isfinite(tr)) return 1.0;
  int e;
  frexp(tr, &e);            // tr = m * 2^e, m in [0.5, 1)
  return ldexp(1.0, -e);    // exact power of two
}

// C = s^2 * M M^T for row-major M (n x k, leading dim ld); s is the power-of-two scale derived from
// the trace of the input (step >= 1) or 1 (step 0, the Gram matrix of Z itself).
__global__ __launch_bounds__(kTile* kTile) void syrk_step_kernel(const double* __restrict__ M, int n, int k, int ld,
                                                                double* __restrict__ C, Rank1Ctl* __restrict__ ctl,
                                                                int step, int out_buf) {
  __shared__ double As[kTile][kTile + 1];
  __shared__ double Bs[kTile][kTile + 1];
  __shared__ double diag[kTile];
  __shared__ double s_scale;
  __shared__ int s_done;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int nt = (n + kTile - 1) / kTile;
  if (tx == 0 && ty == 0) {
    int done = 0;
    double scale = 1.0;
    if (step == 0) {
      if (blockIdx.x == 0 && blockIdx.y == 0) { ctl->done = 0; ctl->final_buf = -1; }
    } else {
      done = ctl->done;
      if (!done) {
        // G_s = (sc_{s-1} G_{s-1})^2 with sc_{s-1} the power-of-two scale of tr(G_{s-1}).
        double tr1;
        scale = pow2_scale_from_trace(ctl->trace[step - 1], nt, &tr1);
        if (step >= 2) {
          // tr(G_{s-1}) = sc_{s-2}^2 tr(G_{s-2}^2), so rho = tr(G_{s-2}^2) / tr(G_{s-2})^2 is
          // 1 - 2*(lambda_2/lambda_1) to first order: at 1 - 1e-13 G_{s-2} is rank one to 5e-14
          // and its square G_{s-1} (the input of this step) is converged far below eps.
          double tr0;
          const double sc0 = pow2_scale_from_trace(ctl->trace[step - 2], nt, &tr0);
          const double rho = tr1 / (sc0 * sc0 * tr0 * tr0);
          if (!(tr1 > 0.0) || rho >= 1.0 - 1e-13) done = 1;
        }
        if (done && blockIdx.x == 0 && blockIdx.y == 0) { ctl->done = 1; ctl->final_buf = out_buf ^ 1; }
      }
    }
    s_scale = scale;
    s_done = done;
  }
  __syncthreads();
  if (s_done) return;
  const double scale = s_scale;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  double acc = 0.0;
  for (int kk = 0; kk < k; kk += kTile) {
    const int col = kk + tx;
    As[ty][tx] = (i0 + ty < n && col < k) ? M[(int64_t)(i0 + ty) * ld + col] : 0.0;
    Bs[ty][tx] = (j0 + ty < n && col < k) ? M[(int64_t)(j0 + ty) * ld + col] : 0.0;
    __syncthreads();
#pragma unroll
    for (int l = 0; l < kTile; ++l) acc = fma(As[ty][l], Bs[tx][l], acc);
    __syncthreads();
  }
  acc *= scale * scale;
  if (i0 + ty < n && j0 + tx < n) C[(int64_t)(i0 + ty) * n + (j0 + tx)] = acc;
  if (blockIdx.x == blockIdx.y) {
    if (tx == ty) diag[tx] = (i0 + ty < n) ? acc : 0.0;
    __syncthreads();
    if (tx == 0 && ty == 0) {
      double t = 0.0;
      for (int l = 0; l < kTile; ++l) t += diag[l];
      ctl->trace[step][blockIdx.x] = t;
    }
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ Z, int A, int B, double* __restrict__ Zt) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)A * B) return;
  const int b = (int)(idx / A), a = (int)(idx % A);   // Zt is B x A
  Zt[idx] = Z[(int64_t)a * B + b];
}

// y = Z x (rows of Z over wavefronts) and y = Z^T x (columns over threads), inside one workgroup.
__device__ void gemv_n(const double* __restrict__ Z, int A, int B, const double* x, double* y) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int j = wv; j < A; j += nw) {
    double s = 0.0;
    for (int k = lane; k < B; k += 64) s = fma(Z[(int64_t)j * B + k], x[k], s);
    s = wave_sum(s);
    if (lane == 0) y[j] = s;
  }
  __syncthreads();
}
__device__ void gemv_t(const double* __restrict__ Z, int A, int B, const double* x, double* y) {
  for (int k = threadIdx.x; k < B; k += blockDim.x) {
    double s = 0.0;
    for (int j = 0; j < A; ++j) s = fma(Z[(int64_t)j * B + k], x[j], s);
    y[k] = s;
  }
  __syncthreads();
}
__device__ double vec_normalize(double* v, int n, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s = fma(v[i], v[i], s);
  s = sqrt(block_sum(s, red));
  for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] = v[i] / s;
  __syncthreads();
  return s;
}

// From the converged Gram power G (n x n, on the smaller side of Z) to the singular pair.
__global__ __launch_bounds__(1024) void rank1_finish_kernel(const double* __restrict__ Z, int A, int B,
                                                           const double* __restrict__ buf0, const double* __restrict__ buf1,
                                                           const Rank1Ctl* __restrict__ ctl, int last_buf,
                                                           double* __restrict__ wA, double* __restrict__ wB,
                                                           double* __restrict__ sigma) {
  extern __shared__ double lds[];
  __shared__ double red[3][16];
  __shared__ int s_arg;
  double* sa = lds;                    // A
  double* sb = lds + ((A + 1) & ~1);   // B
  const bool gram_on_rows = (A <= B);
  const int n = gram_on_rows ? A : B;
  const int fb = (ctl->final_buf >= 0) ? ctl->final_buf : last_buf;
  const double* G = fb ? buf1 : buf0;
  // dominant column of G: the one with the largest diagonal entry (first on ties)
  if (threadIdx.x == 0) {
    int arg = 0;
    double best = G[0];
    for (int i = 1; i < n; ++i) { const double d = G[(int64_t)i * n + i]; if (d > best) { best = d; arg = i; } }
    s_arg = arg;
  }
  __syncthreads();
  double* seed = gram_on_rows ? sa : sb;
  for (int i = threadIdx.x; i < n; i += blockDim.x) seed[i] = G[(int64_t)i * n + s_arg];
  __syncthreads();
  vec_normalize(seed, n, red[0]);
  if (gram_on_rows) { gemv_t(Z, A, B, sa, sb); vec_normalize(sb, B, red[1]); }
  // two exact power steps with Z:  wA = Z wB / |.| ; wB = Z^T wA / |.| ; wA = Z wB / sigma
  gemv_n(Z, A, B, sb, sa);
  vec_normalize(sa, A, red[2]);
  gemv_t(Z, A, B, sa, sb);
  vec_normalize(sb, B, red[0]);
  gemv_n(Z, A, B, sb, sa);
  const double sg = vec_normalize(sa, A, red[1]);
  // sign rule: largest-|.| entry of the last mode's vector is positive (first on ties)
  if (threadIdx.x == 0) {
    int arg = 0;
    double best = fabs(sb[0]);
    for (int i = 1; i < B; ++i) { const double d = fabs(sb[i]); if (d > best) { best = d; arg = i; } }
    s_arg = (sb[arg] < 0.0) ? 1 : 0;
  }
  __syncthreads();
  const double sgn = s_arg ? -1.0 : 1.0;
  for (int i = threadIdx.x; i < A; i += blockDim.x) wA[i] = sgn * sa[i];
  for (int i = threadIdx.x; i < B; i += blockDim.x) wB[i] = sgn * sb[i];
  if (threadIdx.x == 0 && sigma) sigma[0] = sg;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_rank1_workspace_bytes(int A, int B) {
  if (A <= 0 || B <= 0) return 0;
  const size_t n = (size_t)(A < B ? A : B);
  return align_up(sizeof(Rank1Ctl), 256) + 2 * align_up(n * n * sizeof(double), 256) +
         align_up((size_t)A * B * sizeof(double), 256);
}

int cmtfpls_rank1_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, int n_squarings,
                      void* ws, size_t ws_bytes, void* stream) {
  if (!Z || !wA || !wB || A <= 0 || B <= 0) { set_error("rank1: bad argument"); return CMTFPLS_EINVAL; }
  const int n = A < B ? A : B;
  if (n > kTile * kMaxTiles) { set_error("rank1: min(A, B) > 1024 unsupported"); return CMTFPLS_EUNSUPPORTED; }
  if ((size_t)((A + 1) & ~1) + (size_t)((B + 1) & ~1) > 8192) { set_error("rank1: A + B > 8192 unsupported"); return CMTFPLS_EUNSUPPORTED; }
  if (n_squarings < 1) n_squarings = 1;
  if (n_squarings > kMaxSteps) n_squarings = kMaxSteps;
  if (!ws || ws_bytes < cmtfpls_rank1_workspace_bytes(A, B)) { set_error("rank1: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  char* p = static_cast<char*>(ws);
  Rank1Ctl* ctl = reinterpret_cast<Rank1Ctl*>(p);
  p += align_up(sizeof(Rank1Ctl), 256);
  double* buf0 = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * n * sizeof(double), 256);
  double* buf1 = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * n * sizeof(double), 256);
  double* Zt = reinterpret_cast<double*>(p);

  const double* M0 = Z;   // n x k with n on the smaller side
  int k0 = B;
  if (A > B) {
    const int64_t tot = (int64_t)A * B;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Z, A, B, Zt);
    M0 = Zt;
    k0 = A;
  }
  const int nt = (n + kTile - 1) / kTile;
  const dim3 grid(nt, nt), block(kTile, kTile);
  // step 0: G_0 = M0 M0^T -> buf0 ; step s: G_s = scale^2 G_{s-1} G_{s-1}^T -> buf[s & 1]
  hipLaunchKernelGGL(syrk_step_kernel, grid, block, 0, st, M0, n, k0, k0, buf0, ctl, 0, 0);
  for (int s = 1; s <= n_squarings; ++s) {
    const double* in = (s & 1) ? buf0 : buf1;
    double* out = (s & 1) ? buf1 : buf0;
    hipLaunchKernelGGL(syrk_step_kernel, grid, block, 0, st, in, n, n, n, out, ctl, s, s & 1);
  }
  const size_t lds = ((size_t)((A + 1) & ~1) + (size_t)((B + 1) & ~1)) * sizeof(double);
  hipLaunchKernelGGL(rank1_finish_kernel, dim3(1), dim3(1024), lds, st, Z, A, B, buf0, buf1, ctl, n_squarings & 1, wA, wB, sigma);
  return check_launch("rank1");
}

}  // extern "C"
