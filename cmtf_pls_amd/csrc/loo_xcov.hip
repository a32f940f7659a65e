// Leave-one-out refits BEYOND the LDS-resident shapes of loo.hip (validate.get_q2y, cmtf_pls/validate.py:7-37; SURVEY 8(f).2
// "down-date S and the means instead of refitting"): all folds of a launch side by side, ONE 1024-thread WORKGROUP PER FOLD
// running that fold's whole tPLS.fit (tpls.py:73-113) and the prediction of its held-out sample (tpls.py:122-143), for
// trailing shapes up to min(A, B) = 256 (128 x 128, 256 x 256, ...), where loo.hip's per-fold vectors (Z, two n x n Gram
// buffers) no longer fit the LDS and the product used to refit once per fold on the regular engine.
//
// What a fold does NOT recompute or re-read:
//  * the means: (column sums of all samples - the held-out row) / (I - 1); the held-out row is zero in the fold's centred
//    working copies, which removes it from every sum (as loo.hip);
//  * the tensor inside the NIPALS loop: within a component X_f and Y_f are fixed and u = Y_f q, so
//        np.einsum(X, u) = S^T q   (tpls.py:83),   Y.T @ t = S (wA (x) wB)   (tpls.py:100),   |u_old - u|^2 = dq^T (Y_f^T Y_f) dq   (tpls.py:103)
//    with S = Y_f^T X_f (M x P) formed ONCE per component: an inner iteration reads the 2 M P doubles of S instead of the
//    2 I P of the fold's tensor (I / M times less: 32 x at 512 samples, 16 responses) -- the cross-covariance re-association of
//    the engine's algorithm="xcov", here inside one workgroup.  Per component the fold's tensor is read for S, for the final
//    score, and read + written by the deflation (tpls.py:109).
//  * the rank-1 extraction (tpls.py:86-88) is the product's: Gram matrix of the smaller side squared repeatedly with
//    power-of-two rescaling until numerically rank one, each n x n x n product on the f64 matrix cores
//    (v_mfma_f64_16x16x4_f64, operands straight from L2 in the MFMA layout, the 16 wavefronts of the workgroup dealing the
//    lower-triangular 16 x 16 tiles among themselves), one exact pass with Z, sign rule on the last mode.
// Arithmetic: float64 throughout, the reference's operation order outside the re-association above.
// Limits: X of order 2 or 3 without missing values, min(A, B) <= 256, M <= 128, R <= 64, the workgroup's small vectors in
// 150 KB of LDS; per resident fold a workspace of I P + M P + 3 P + 2 n^2 + I (M + R + 2) + R (A + B) doubles (cmtfpls_loo_xcov_fold_workspace_bytes).
#include "common.hpp"

namespace cmtfpls {

constexpr int kLxNT = 1024, kLxWaves = kLxNT / 64;
constexpr int kLxMaxN = 256, kLxMaxM = 128, kLxMaxR = 64;     // (M: as far as the M x M Gram of the responses fits the LDS next to the rest)

typedef double lx_d4_t __attribute__((ext_vector_type(4)));

struct LooXArgs {
  const double* X;        // (I, P) original, uncentred
  const double* Y;        // (I, M)
  const double* colsum_x; // (P)
  const double* colsum_y; // (M)
  double* ws;             // per resident fold, see carve-up in the kernel
  double* Ypred;          // (I, M): row i = prediction of the model fitted without sample i
  int* n_iter;            // (I, R), nullable
  int64_t ws_per_fold;    // doubles
  int I, A, B, M, R, max_iter, fold0, nfolds;
  double tol;
};

// sum over the workgroup; every thread gets the same value; two barriers, so back-to-back calls may share `red`
__device__ __forceinline__ double lx_sum(double v, double* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < kLxWaves; ++w) s += red[w];
  __syncthreads();
  return s;
}

// C (n x n) = scale2 * Mx Mx^T for row-major Mx (n x k, leading dimension ld), C_keep (nullable) a second copy.
// Lower-triangular 16 x 16 tiles dealt round-robin to the 16 wavefronts, each on the f64 matrix cores:
//   v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][kq = l >> 4] and B[kq][j = l & 15] and holds D[(l >> 4) + 4 e][l & 15];
//   lane group kq takes the 8 consecutive columns c0 + 8 kq + (0..7) of a 32-column chunk, MFMA s multiplies column
//   c0 + 8 kq + s of row i0 + (l & 15) with the same column of row j0 + (l & 15) (B = Mx^T).
// The mirrored tile is written from the same registers (C is bitwise symmetric).  Returns tr(C) and |C|_F^2 to every thread.
__device__ void lx_syrk(const double* Mx, int n, int k, int ld, double* C, double* C_keep, double scale2, double* red,
                        double* tr_out, double* fro_out) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ri = lane & 15, kq = lane >> 4;
  const int nt = (n + 15) / 16;
  double trp = 0.0, frp = 0.0;
  int idx = 0;
  for (int ti = 0; ti < nt; ++ti)
    for (int tj = 0; tj <= ti; ++tj, ++idx) {
      if ((idx % kLxWaves) != wv) continue;
      const int i0 = ti * 16, j0 = tj * 16;
      const bool ra = (i0 + ri) < n, rb = (j0 + ri) < n;
      const double* rowa = Mx + (int64_t)(ra ? i0 + ri : 0) * ld;
      const double* rowb = Mx + (int64_t)(rb ? j0 + ri : 0) * ld;
      lx_d4_t acc = lx_d4_t{0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < k; kk += 32) {
        double a[8], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int c = kk + 8 * kq + s;
          const int cc = (c < k) ? c : 0;
          a[s] = rowa[cc];
          b[s] = rowb[cc];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bool cok = (kk + 8 * kq + s) < k;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ra && cok) ? a[s] : 0.0, (rb && cok) ? b[s] : 0.0, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = i0 + kq + 4 * e, c = j0 + ri;
        if (r < n && c < n) {
          const double v = acc[e] * scale2;
          C[(int64_t)r * n + c] = v;
          if (C_keep) C_keep[(int64_t)r * n + c] = v;
          if (ti != tj) {
            C[(int64_t)c * n + r] = v;
            if (C_keep) C_keep[(int64_t)c * n + r] = v;
            frp = fma(2.0 * v, v, frp);
          } else {
            frp = fma(v, v, frp);
            if (r == c) trp += v;
          }
        }
      }
    }
  *tr_out = lx_sum(trp, red);        // (the barriers inside also publish C to the whole workgroup)
  *fro_out = lx_sum(frp, red);
}

// Leading singular pair of Z (A x B row-major, global): wA (A), wB (B) unit norm, largest-|.| entry of wB positive.
// Zt: P doubles of scratch (the transpose when B < A); G0 / G1: n x n each (ping-pong); xs (n), ys (k) in LDS.
__device__ void lx_rank1(const double* Z, double* Zt, int A, int B, double* wA, double* wB, double* G0, double* G1,
                         double* xs, double* ys, double* red, double* bestv, int* besti) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool rowsA = A <= B;                        // Mx = Z (n = A) or Z^T (n = B)
  const int n = rowsA ? A : B, k = rowsA ? B : A;
  const double* Mx = Z;
  if (!rowsA) {
    for (int idx = tid; idx < A * B; idx += kLxNT) { const int b = idx / A, a = idx % A; Zt[idx] = Z[(int64_t)a * B + b]; }
    __syncthreads();
    Mx = Zt;
  }
  double tr, fro;
  lx_syrk(Mx, n, k, k, G0, nullptr, 1.0, red, &tr, &fro);                              // G_0 = Mx Mx^T
  double* G = G0;
  double* Gn = G1;
  for (int step = 0; step < 64; ++step) {
    if (!(tr > 0.0) || !isfinite(tr) || fro / (tr * tr) >= 1.0 - 1e-13) break;         // uniform: numerically rank one
    int e;
    frexp(tr, &e);
    const double sc = ldexp(1.0, -e);                                                 // exact power of two
    lx_syrk(G, n, n, n, Gn, nullptr, sc * sc, red, &tr, &fro);                         // G <- (sc G)^2   (G symmetric: G G = G G^T)
    double* tmp = G; G = Gn; Gn = tmp;
  }
  // seed = dominant column of G (first index on ties), normalised
  double bv = -1.0;
  int bi = 0;
  for (int i = tid; i < n; i += kLxNT) { const double d = G[(int64_t)i * n + i]; if (d > bv) { bv = d; bi = i; } }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ov = __shfl_xor(bv, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { bestv[wv] = bv; besti[wv] = bi; }
  __syncthreads();
  bv = bestv[0];
  bi = besti[0];
  for (int w = 1; w < kLxWaves; ++w)
    if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
  __syncthreads();
  double ss = 0.0;
  for (int i = tid; i < n; i += kLxNT) { const double g = G[(int64_t)bi * n + i]; ss = fma(g, g, ss); }
  const double snrm = sqrt(lx_sum(ss, red));
  for (int i = tid; i < n; i += kLxNT) xs[i] = G[(int64_t)bi * n + i] / snrm;
  __syncthreads();
  for (int l = tid; l < k; l += kLxNT) {                                               // y = Mx^T seed
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = fma(Mx[(int64_t)i * k + l], xs[i], s);
    ys[l] = s;
  }
  __syncthreads();
  for (int i = wv; i < n; i += kLxWaves) {                                             // x = Mx y: a wavefront per row
    double s = 0.0;
    for (int l = lane; l < k; l += 64) s = fma(Mx[(int64_t)i * k + l], ys[l], s);
    s = wave_sum(s);
    if (lane == 0) xs[i] = s;                 // (the seed is dead: every wavefront finished y before the barrier above)
  }
  __syncthreads();
  double sx = 0.0, sy = 0.0;
  for (int i = tid; i < n; i += kLxNT) sx = fma(xs[i], xs[i], sx);
  for (int l = tid; l < k; l += kLxNT) sy = fma(ys[l], ys[l], sy);
  const double nx = sqrt(lx_sum(sx, red)), ny = sqrt(lx_sum(sy, red));
  // sign rule on the LAST mode's vector wB: its largest-|.| entry is positive (first index on ties)
  const double* vb = rowsA ? ys : xs;
  const int nb = rowsA ? k : n;
  bv = -1.0;
  bi = 0;
  for (int i = tid; i < nb; i += kLxNT) { const double d = fabs(vb[i]); if (d > bv) { bv = d; bi = i; } }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ov = __shfl_xor(bv, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { bestv[wv] = bv; besti[wv] = bi; }
  __syncthreads();
  bv = bestv[0];
  bi = besti[0];
  for (int w = 1; w < kLxWaves; ++w)
    if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
  const double sgn = (vb[bi] < 0.0) ? -1.0 : 1.0;
  double* ox = rowsA ? wA : wB;
  double* oy = rowsA ? wB : wA;
  __syncthreads();
  for (int i = tid; i < n; i += kLxNT) ox[i] = sgn * (xs[i] / nx);
  for (int l = tid; l < k; l += kLxNT) oy[l] = sgn * (ys[l] / ny);
  __syncthreads();
}

// sum_c row[c] * wk[c] over one wavefront's columns c = lane, lane + 64, ...: one fma chain per lane in column order (the order of the
// plain loop), eight loads of each operand in flight per trip
__device__ __forceinline__ double lx_wave_dot(const double* row, const double* wk, int64_t P, int lane) {
  double s = 0.0;
  int64_t c = lane;
  for (; c + 7 * 64 < P; c += 8 * 64) {
    double xv[8], wv8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xv[j] = row[c + 64 * j]; wv8[j] = wk[c + 64 * j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fma(xv[j], wv8[j], s);
  }
  for (; c < P; c += 64) s = fma(row[c], wk[c], s);
  return wave_sum(s);
}

__global__ __launch_bounds__(kLxNT) void loo_xcov_kernel(LooXArgs a) {
  extern __shared__ double sm[];
  __shared__ double red[kLxWaves];
  __shared__ double bestv[kLxWaves];
  __shared__ int besti[kLxWaves];
  __shared__ double scv[kLxMaxR];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int I = a.I, A = a.A, B = a.B, M = a.M, R = a.R;
  const int64_t P = (int64_t)A * B;
  const int n = A < B ? A : B, k = A < B ? B : A;
  if ((int)blockIdx.x >= a.nfolds) return;
  const int fold = a.fold0 + blockIdx.x;
  if (fold >= I) return;
  // global carve-up of this fold's workspace
  double* Xf = a.ws + (int64_t)blockIdx.x * a.ws_per_fold;   // I x P   centred, held-out row zero, deflated in place
  double* Yf = Xf + (int64_t)I * P;                          // I x M
  double* T = Yf + (int64_t)I * M;                           // I x R
  double* S = T + (int64_t)I * R;                            // M x P   cross-covariance of the current component
  double* Z = S + (int64_t)M * P;                            // P
  double* Zt = Z + P;                                        // P       (transpose scratch of the rank-1 extraction)
  double* wk = Zt + P;                                       // P       kron(wA, wB) of the current loadings
  double* G0 = wk + P;                                       // n x n
  double* G1 = G0 + (int64_t)n * n;
  double* u = G1 + (int64_t)n * n;                           // I
  double* t = u + I;                                         // I
  double* Wa = t + I;                                        // R x A   loadings of the components so far (read again by the prediction only)
  double* Wb = Wa + (int64_t)R * A;                          // R x B
  // LDS carve-up
  double* wA = sm;
  double* wB = wA + A;
  double* q = wB + B;
  double* qn = q + M;
  double* tq = qn + M;
  double* my = tq + M;
  double* Gy = my + M;            // M x M   Y_f^T Y_f of the current component
  double* xs = Gy + M * M;        // n
  double* ys = xs + n;            // k
  double* coef = ys + k;          // R x R
  double* Qs = coef + R * R;      // R x M
  double* Gn = Qs + R * M;        // (a+1) x (a+1) normal equations
  double* gn = Gn + R * R;
  double* bb = gn + R;
  double* dd = bb + R;
  const double inv = 1.0 / (double)(I - 1);

  // ---- preprocess (tpls.py:61-71): the fold's means by down-dating the column sums; centred copies with the held-out row zero
  for (int o = tid; o < R * R; o += kLxNT) coef[o] = 0.0;
  for (int m = tid; m < M; m += kLxNT) my[m] = (a.colsum_y[m] - a.Y[(int64_t)fold * M + m]) * inv;
  for (int64_t c = tid; c < P; c += kLxNT) Z[c] = (a.colsum_x[c] - a.X[(int64_t)fold * P + c]) * inv;
  __syncthreads();
  for (int r = 0; r < I; ++r) {
    const double* xr = a.X + (int64_t)r * P;
    double* xo = Xf + (int64_t)r * P;
    if (r == fold) { for (int64_t c = tid; c < P; c += kLxNT) xo[c] = 0.0; }
    else { for (int64_t c = tid; c < P; c += kLxNT) xo[c] = xr[c] - Z[c]; }
  }
  for (int64_t idx = tid; idx < (int64_t)I * M; idx += kLxNT) {
    const int r = (int)(idx / M), m = (int)(idx % M);
    Yf[idx] = (r == fold) ? 0.0 : a.Y[idx] - my[m];
  }
  __syncthreads();

  for (int comp = 0; comp < R; ++comp) {
    // ---- S = Y_f^T X_f and G_y = Y_f^T Y_f of this component (X_f, Y_f as deflated so far) ----
    for (int64_t c = tid; c < P; c += kLxNT) {
      for (int mc = 0; mc < M; mc += 16) {
        double acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0;
        // four rows of the column in flight at a time (one row per trip leaves a single load per lane outstanding: the pass
        // is then bound by memory latency, 16 ms per component and fold at 512 x 16384 instead of 2)
        int r = 0;
        for (; r + 4 <= I; r += 4) {
          double x[4];
#pragma unroll
          for (int u4 = 0; u4 < 4; ++u4) x[u4] = Xf[(int64_t)(r + u4) * P + c];
#pragma unroll
          for (int u4 = 0; u4 < 4; ++u4) {
            const double* yr = Yf + (int64_t)(r + u4) * M + mc;
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (mc + j < M) acc[j] = fma(yr[j], x[u4], acc[j]);
          }
        }
        for (; r < I; ++r) {
          const double x = Xf[(int64_t)r * P + c];
          const double* yr = Yf + (int64_t)r * M + mc;
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (mc + j < M) acc[j] = fma(yr[j], x, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (mc + j < M) S[(int64_t)(mc + j) * P + c] = acc[j];
      }
    }
    for (int o = tid; o < M * M; o += kLxNT) {
      const int m1 = o / M, m2 = o % M;
      double s = 0.0;
      for (int r = 0; r < I; ++r) s = fma(Yf[(int64_t)r * M + m1], Yf[(int64_t)r * M + m2], s);
      Gy[o] = s;
    }
    for (int m = tid; m < M; m += kLxNT) q[m] = (m == 0) ? 1.0 : 0.0;                 // u_0 = Y_f[:, 0] = Y_f e_0 (tpls.py:78)
    __syncthreads();
    int it = 0;
    for (; it < a.max_iter; ++it) {                                                    // tpls.py:79
      for (int64_t c = tid; c < P; c += kLxNT) {                                       // Z = X x_0 u = S^T q (tpls.py:83)
        double s = 0.0;
        int m = 0;
        for (; m + 4 <= M; m += 4) {                                                   // (four rows of S in flight; same order of the sum)
          const double s0 = S[(int64_t)m * P + c], s1 = S[(int64_t)(m + 1) * P + c], s2 = S[(int64_t)(m + 2) * P + c], s3 = S[(int64_t)(m + 3) * P + c];
          s = fma(q[m], s0, s);
          s = fma(q[m + 1], s1, s);
          s = fma(q[m + 2], s2, s);
          s = fma(q[m + 3], s3, s);
        }
        for (; m < M; ++m) s = fma(q[m], S[(int64_t)m * P + c], s);
        Z[c] = s;
      }
      __syncthreads();
      if (A == 1) {                                                                    // tpls.py:84: Z / norm(Z)
        double s = 0.0;
        for (int64_t c = tid; c < P; c += kLxNT) s = fma(Z[c], Z[c], s);
        const double nz = sqrt(lx_sum(s, red));
        for (int64_t c = tid; c < P; c += kLxNT) wB[c] = Z[c] / nz;
        if (tid == 0) wA[0] = 1.0;
        __syncthreads();
      } else {
        lx_rank1(Z, Zt, A, B, wA, wB, G0, G1, xs, ys, red, bestv, besti);          // tpls.py:86-88
      }
      for (int64_t c = tid; c < P; c += kLxNT) wk[c] = wA[c / B] * wB[c % B];          // the Kronecker loading, once per extraction
      __syncthreads();
      for (int m = wv; m < M; m += kLxWaves) {                                         // Y^T t = S (wA (x) wB) (tpls.py:97-100)
        const double s = lx_wave_dot(S + (int64_t)m * P, wk, P, lane);
        if (lane == 0) tq[m] = s;
      }
      __syncthreads();
      double qs = 0.0;
      for (int m = tid; m < M; m += kLxNT) qs = fma(tq[m], tq[m], qs);
      const double qnrm = sqrt(lx_sum(qs, red));
      for (int m = tid; m < M; m += kLxNT) qn[m] = tq[m] / qnrm;                       // tpls.py:101
      __syncthreads();
      double d2 = 0.0;                                                                 // |u_old - u|^2 = dq^T G_y dq (tpls.py:102-103)
      for (int o = tid; o < M * M; o += kLxNT) d2 = fma((qn[o / M] - q[o / M]) * Gy[o], qn[o % M] - q[o % M], d2);
      d2 = lx_sum(d2, red);
      for (int m = tid; m < M; m += kLxNT) q[m] = qn[m];
      __syncthreads();
      if (it > 0 && sqrt(d2 > 0.0 ? d2 : 0.0) < a.tol) { ++it; break; }              // first pass: oldU = inf (tpls.py:77)
    }
    if (a.n_iter && tid == 0) a.n_iter[(int64_t)fold * R + comp] = it;
    // ---- the component's score and Y score with the converged loadings (tpls.py:97-102) ----
    for (int r = wv; r < I; r += kLxWaves) {                                             // (wk holds the converged loadings' Kronecker product)
      const double s = lx_wave_dot(Xf + (int64_t)r * P, wk, P, lane);
      if (lane == 0) t[r] = s;
    }
    for (int r = tid; r < I; r += kLxNT) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) s = fma(Yf[(int64_t)r * M + m], q[m], s);
      u[r] = s;
    }
    __syncthreads();
    for (int r = tid; r < I; r += kLxNT) T[(int64_t)r * R + comp] = t[r];
    for (int j = tid; j < A; j += kLxNT) Wa[comp * A + j] = wA[j];
    for (int j = tid; j < B; j += kLxNT) Wb[comp * B + j] = wB[j];
    for (int m = tid; m < M; m += kLxNT) Qs[comp * M + m] = q[m];
    // ---- deflate X (tpls.py:109) ----
    for (int r = 0; r < I; ++r) {
      const double tr = t[r];
      double* xr = Xf + (int64_t)r * P;
      for (int64_t c = tid; c < P; c += kLxNT) xr[c] = fma(-tr, wk[c], xr[c]);
    }
    __syncthreads();
    // ---- inner regression b = lstsq(T[:, :k], u) (tpls.py:110-112): normal equations, equilibrated Cholesky (as loo.hip) ----
    const int kk = comp + 1;
    for (int o = tid; o < kk * kk + kk; o += kLxNT) {
      double s = 0.0;
      if (o < kk * kk) {
        const int p = o / kk, s2 = o % kk;
        for (int r = 0; r < I; ++r) s = fma(T[(int64_t)r * R + p], T[(int64_t)r * R + s2], s);
        Gn[o] = s;
      } else {
        const int p = o - kk * kk;
        for (int r = 0; r < I; ++r) s = fma(T[(int64_t)r * R + p], u[r], s);
        gn[p] = s;
      }
    }
    __syncthreads();
    if (tid == 0) {
      const double tiny = (double)kk * 2.220446049250313e-16;
      for (int i = 0; i < kk; ++i) { const double g = Gn[i * kk + i]; dd[i] = (g > 0.0 && isfinite(g)) ? 1.0 / sqrt(g) : 0.0; }
      for (int i = 0; i < kk; ++i) {
        for (int j = 0; j < kk; ++j) Gn[i * kk + j] *= dd[i] * dd[j];
        bb[i] = gn[i] * dd[i];
      }
      bool dep[kLxMaxR];
      for (int c = 0; c < kk; ++c) {
        const double piv = Gn[c * kk + c];
        dep[c] = !(piv > tiny);
        if (dep[c]) { Gn[c * kk + c] = 1.0; for (int i = c + 1; i < kk; ++i) Gn[i * kk + c] = 0.0; continue; }
        const double l = sqrt(piv);
        Gn[c * kk + c] = l;
        for (int i = c + 1; i < kk; ++i) Gn[i * kk + c] /= l;
        for (int i = c + 1; i < kk; ++i)
          for (int j = c + 1; j <= i; ++j) Gn[i * kk + j] -= Gn[i * kk + c] * Gn[j * kk + c];
      }
      for (int r = 0; r < kk; ++r) {
        double s = bb[r];
        for (int j = 0; j < r; ++j) s -= Gn[r * kk + j] * bb[j];
        bb[r] = dep[r] ? 0.0 : s / Gn[r * kk + r];
      }
      for (int r = kk - 1; r >= 0; --r) {
        double s = bb[r];
        for (int j = r + 1; j < kk; ++j) s -= Gn[j * kk + r] * bb[j];
        bb[r] = dep[r] ? 0.0 : s / Gn[r * kk + r];
      }
      for (int r = 0; r < kk; ++r) { bb[r] *= dd[r]; coef[r * R + comp] = bb[r]; }
    }
    __syncthreads();
    // ---- Y -= T b q^T (tpls.py:113); t is free: reuse it for yhat = T b ----
    for (int r = tid; r < I; r += kLxNT) {
      double s = 0.0;
      for (int j = 0; j < kk; ++j) s = fma(T[(int64_t)r * R + j], bb[j], s);
      t[r] = s;
    }
    __syncthreads();
    for (int64_t idx = tid; idx < (int64_t)I * M; idx += kLxNT) {
      const int r = (int)(idx / M), m = (int)(idx % M);
      Yf[idx] = fma(-t[r], q[m], Yf[idx]);
    }
    __syncthreads();
  }

  // ---- predict the held-out sample (tpls.py:122-143): centre with the fold's means, project and deflate ----
  for (int64_t c = tid; c < P; c += kLxNT) {
    const double xv = a.X[(int64_t)fold * P + c];
    Z[c] = xv - (a.colsum_x[c] - xv) * inv;
  }
  __syncthreads();
  double* sc = scv;                                                                     // scores of the held-out row (R)
  for (int comp = 0; comp < R; ++comp) {
    double s = 0.0;
    for (int64_t c = tid; c < P; c += kLxNT) s = fma(Z[c], Wa[comp * A + c / B] * Wb[comp * B + c % B], s);
    const double sv = lx_sum(s, red);
    if (tid == 0) sc[comp] = sv;
    for (int64_t c = tid; c < P; c += kLxNT) Z[c] = fma(-sv, Wa[comp * A + c / B] * Wb[comp * B + c % B], Z[c]);
    __syncthreads();
  }
  for (int m = tid; m < M; m += kLxNT) {
    double yv = 0.0;
    for (int b2 = 0; b2 < R; ++b2) {
      double sb = 0.0;
      for (int a2 = 0; a2 < R; ++a2) sb = fma(sc[a2], coef[a2 * R + b2], sb);        // (scores @ coef_)[b]
      yv = fma(sb, Qs[b2 * M + m], yv);                                             // @ Q^T
    }
    a.Ypred[(int64_t)fold * M + m] = yv + my[m];
  }
}

static size_t lx_lds_bytes(int A, int B, int M, int R) {
  const size_t n = (size_t)(A < B ? A : B), k = (size_t)(A < B ? B : A);
  const size_t dbl = (size_t)A + B + 4 * (size_t)M + (size_t)M * M + n + k + (size_t)R * R + (size_t)R * M + (size_t)R * R + 3 * (size_t)R;
  return dbl * sizeof(double);
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_loo_xcov_fold_workspace_bytes(int I, int A, int B, int M, int R) {
  if (I <= 1 || A <= 0 || B <= 0 || M <= 0 || R <= 0) return 0;
  const size_t P = (size_t)A * B, n = (size_t)(A < B ? A : B);
  return ((size_t)I * P + (size_t)I * M + (size_t)I * R + (size_t)M * P + 3 * P + 2 * n * n + 2 * (size_t)I + (size_t)R * ((size_t)A + B)) * sizeof(double);
}

int cmtfpls_loo_xcov_f64(const double* X, const double* Y, const double* colsum_x, const double* colsum_y, int I, int A, int B, int M,
                         int R, double tol, int max_iter, int fold0, int nfolds, double* Ypred, int* n_iter, void* ws,
                         size_t ws_bytes, void* stream) {
  if (!X || !Y || !colsum_x || !colsum_y || !Ypred || I <= 1 || A <= 0 || B <= 0 || M <= 0 || R <= 0 || max_iter <= 0 || fold0 < 0 ||
      nfolds <= 0 || fold0 + nfolds > I) {
    set_error("loo_xcov: bad argument");
    return CMTFPLS_EINVAL;
  }
  const int n = A < B ? A : B;
  const size_t lds = lx_lds_bytes(A, B, M, R);
  if (n > kLxMaxN || M > kLxMaxM || R > kLxMaxR || lds > 150 * 1024 || (int64_t)A * B > (int64_t)1 << 24) {
    set_error("loo_xcov: shape outside the workgroup-per-fold form (min(A, B) <= 256, M <= 128, R <= 64, small vectors within 150 KB of LDS); refit per fold on the regular engine");
    return CMTFPLS_EUNSUPPORTED;
  }
  const size_t per = cmtfpls_loo_xcov_fold_workspace_bytes(I, A, B, M, R);
  if (!ws || ws_bytes < per * (size_t)nfolds) { set_error("loo_xcov: workspace too small"); return CMTFPLS_EWORKSPACE; }
  LooXArgs a;
  a.X = X; a.Y = Y; a.colsum_x = colsum_x; a.colsum_y = colsum_y; a.ws = static_cast<double*>(ws); a.Ypred = Ypred; a.n_iter = n_iter;
  a.ws_per_fold = (int64_t)(per / sizeof(double));
  a.I = I; a.A = A; a.B = B; a.M = M; a.R = R; a.max_iter = max_iter; a.fold0 = fold0; a.nfolds = nfolds; a.tol = tol;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loo_xcov_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(loo_xcov_kernel, dim3(nfolds), dim3(kLxNT), lds, (hipStream_t)stream, a);
  return check_launch("loo_xcov");
}

}  // extern "C"
