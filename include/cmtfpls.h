/*
 * cmtfpls.h -- C ABI of the MI355X (gfx950) NIPALS engine for tensor PLS / coupled tensor PLS.
 *
 * The reference (meyer-lab/cmtf-pls) has no FFI seam: its hot path is a chain of NumPy / tensorly
 * calls inside tPLS.fit (cmtf_pls/tpls.py:73-120) and ctPLS.fit (cmtf_pls/cmtf.py:85-140).  Each
 * entry point below replaces ONE of those call sites (cited per function) and is what a binding on
 * the reference side would call (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's data_ptr());
 *   - X is C-order (I, P) where P is the product of all trailing modes: the mode-0 unfolding is a
 *     free view, every sweep reads contiguous rows;  X is stored as f32 or f64 (suffix), every
 *     other operand (scores, loadings, Y, sums) is f64;
 *   - trailing-mode loadings are passed as two vectors wA (length A) and wB (length B), A*B == P,
 *     meaning w[c] = wA[c / B] * wB[c % B]  (order-3 X: wA = w_J, wB = w_K; a matrix X: A = 1);
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it; nothing here
 *     allocates, synchronises or throws: scratch comes from the caller (`ws`, size from the
 *     matching *_workspace_bytes);
 *   - return value: 0 ok, 1 bad argument, 2 workspace too small, 3 HIP error, 4 unsupported shape.
 *   - missing values are NaNs stored in-band in X (they persist through deflation exactly as in
 *     the reference, where NaN - x = NaN).
 */
#ifndef CMTFPLS_H
#define CMTFPLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMTFPLS_OK 0
#define CMTFPLS_EINVAL 1
#define CMTFPLS_EWORKSPACE 2
#define CMTFPLS_EHIP 3
#define CMTFPLS_EUNSUPPORTED 4

int cmtfpls_abi_version(void);
const char* cmtfpls_last_error(void);
/* Reset the HIP runtime's per-thread last error and this library's message; returns 1 if an error was pending.  For callers
 * that capture launch sequences into HIP graphs: a capture that fails leaves the error set, and the next entry would report it. */
int cmtfpls_clear_error(void);
/* A few status words (convergence norm, rank-1 flags) to PINNED host memory behind the work enqueued so far, then `event`
 * (a hipEvent_t, nullable) recorded: what the host waits on once per NIPALS iteration (tpls.py:103-107) -- one call instead
 * of a framework copy + record, because the pipelined inner loop (engine.FitRun._inner_loop_xcov_pipelined) is bound by host time. */
int cmtfpls_status_to_host(const void* src, void* dst_host, size_t bytes, void* event, void* stream);

/* ---- preprocess: tpls.py:61-71, cmtf.py:74-83 ------------------------------------------------
 * colstats: colsum[c] = sum over non-NaN i of X[i,c]; colcnt[c] = number of non-NaN i
 *           (np.nanmean = colsum / colcnt; np.isnan mask counts).  ws >= colstats_workspace_bytes.
 * center:   X[i,c] -= mean[c] in place (NaN stays NaN); rowcnt[i] (nullable) = non-NaN count of
 *           row i; ssq_part (nullable, >= center_partials() doubles) = per-block partial sums of
 *           the squared centred observed entries (||X_c||^2, denominator of calcR2X util.py:14). */
size_t cmtfpls_colstats_workspace_bytes(int64_t I, int64_t P);
int cmtfpls_colstats_f32(const float* X, int64_t I, int64_t P, double* colsum, double* colcnt,
                         void* ws, size_t ws_bytes, void* stream);
int cmtfpls_colstats_f64(const double* X, int64_t I, int64_t P, double* colsum, double* colcnt,
                         void* ws, size_t ws_bytes, void* stream);
int cmtfpls_sweep_partials(void); /* number of doubles every *ssq_part* argument must hold */
int cmtfpls_center_f32(float* X, int64_t I, int64_t P, const double* mean, double* rowcnt,
                       double* ssq_part, void* stream);
int cmtfpls_center_f64(double* X, int64_t I, int64_t P, const double* mean, double* rowcnt,
                       double* ssq_part, void* stream);

/* ---- K1 mode-0 contraction: np.einsum("i...,i...->...", X, u)  tpls.py:83, cmtf.py:94 --------
 * Z[c] = sum_i X[i,c] * u[i]   (f64 accumulation, deterministic two-stage sum).
 * masked != 0: NaN entries contribute 0 (numerator of miss_tensordot, missingvals.py:19);
 * the I/n_obs rescale is cmtfpls_colscale_f64 so that it can follow a cross-GPU all-reduce. */
size_t cmtfpls_mode0_contract_workspace_bytes(int64_t I, int64_t P);
int cmtfpls_mode0_contract_f32(const float* X, int64_t I, int64_t P, const double* u, double* Z,
                               int masked, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_mode0_contract_f64(const double* X, int64_t I, int64_t P, const double* u, double* Z,
                               int masked, void* ws, size_t ws_bytes, void* stream);
/* The same contraction with u = Y q (tpls.py:102, `Y @ Y_load`) formed inside the kernel instead of
 * being read: Z[c] = sum_i X[i,c] * (Y[i,:] . q).  Saves the u = Y q launch of every iteration.
 * Vector shapes (P % (16/sizeof(T)) == 0) with M <= 64 only: CMTFPLS_EUNSUPPORTED otherwise (form u with
 * cmtfpls_rowdot_f64 and call cmtfpls_mode0_contract_*).  Same workspace as mode0_contract. */
int cmtfpls_mode0_contract_yq_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M,
                                  const double* q, double* Z, int masked, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_mode0_contract_yq_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M,
                                  const double* q, double* Z, int masked, void* ws, size_t ws_bytes, void* stream);
/* Z[c] = colcnt[c] > 0 ? Z[c] / colcnt[c] * n_samples : 0     (missingvals.py:17-19) */
int cmtfpls_colscale_f64(double* Z, int64_t P, const double* colcnt, double n_samples, void* stream);

/* ---- K2 rank-1 extraction ----------------------------------------------------------------------
 * rank1: leading singular pair of the A x B matrix Z: what
 *   parafac(Z, 1, tol, init="svd", normalize_factors=True)[1]   (tpls.py:86-88, cmtf.py:100-102)
 * returns for a matrix Z.  wA = u1, wB = v1 (unit norm), sigma[0] = sigma_1 (nullable); sign: the
 * largest-|.| entry of wB is positive, wA follows (sigma > 0).  Method: repeated squaring of the Gram
 * matrix of the smaller side (at most n_squarings launches, the ones after convergence return at
 * once), then one exact pass y = M^T seed, x = M y with Z itself.  info (nullable, 2 doubles):
 * info[0] = 1 if the squaring was seen to converge within n_squarings (else the caller should call
 * again with a larger budget), info[1] = squarings actually computed (budget hint for the next call).
 * Limit: min(A, B) <= 4096 (round 2: 1024), CMTFPLS_EUNSUPPORTED beyond.
 * Round 4: for min(A, B) <= 256 and n_squarings <= 31 the whole chain of squarings is ONE launch (its (n/16)^2 workgroups stay
 * resident and pass the 16-row panels of G_s to each other through 8-byte agent-scope stores whose value is their own flag, no
 * grid barrier, no fence); cmtfpls_rank1_launches_f64 always takes the launch-per-squaring form (same bits: what the tests
 * compare the chain with).
 * normalize: v /= ||v||_2, the vector case `Z / norm(Z)` (tpls.py:84, cmtf.py:98) and
 * `q /= norm(q)` (tpls.py:101); nrm (nullable) receives the norm. */
size_t cmtfpls_rank1_workspace_bytes(int A, int B);
int cmtfpls_rank1_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                      int n_squarings, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_rank1_launches_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                               int n_squarings, void* ws, size_t ws_bytes, void* stream);
/* The one-launch chain needs all of its workgroups resident at once: true on a GPU the process has to itself.  If a workgroup
 * waits in vain (tens of milliseconds: several processes sharing the card) the extraction returns NaN loadings and info =
 * [0, -1]; the caller then switches the chain off for the process (every entry that extracts a rank-1 pair takes the launch
 * form from then on) and repeats the call.  on = 2 (tests only): the chain runs with one row of its workgroups missing, which
 * exercises exactly that path. */
void cmtfpls_rank1_chain_enable(int on);
int cmtfpls_rank1_chain_enabled(void);
/* rank1 followed by the score of the M rows of S with the loading just formed, tq[m] = S[m,:] . (wA (x) wB) -- the pair every
 * iteration of the cross-covariance loop issues (tpls.py:84-90, then Y^T t = S w) -- with the extraction's last kernel and the
 * score in ONE launch when S has M <= 64 rows of >= 8192 elements (B even); any other shape runs the two entries one after the
 * other.  Same results as cmtfpls_rank1_f64 + cmtfpls_score_f64, bit for bit. */
int cmtfpls_rank1_score_f64(const double* Z, int A, int B, double* wA, double* wB, double* info, int n_squarings,
                            const double* S, int M, double* tq, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_normalize_f64(double* v, int64_t n, double* nrm, void* stream);
/* rank1_tensor: the same parafac call when Z is a TENSOR of order n = 3 .. 7 (X of order 4 .. 8; orders 4 and 5 are
 * exercised by tests/test_cmtf.py:18-21, tests/test_tpls.py:132-155): leading-left-singular-vector
 * init of every unfolding, ALS sweeps, stop when |d rec_error| < tol from the 2nd sweep on (<= 100
 * sweeps), as tensorly 0.9.0 publishes it.  dims: HOST array of the n mode sizes (each <= 1024);
 * factors: device (n x ld) row-major, row m = factor of mode m; info as for rank1 (info[1] = sweeps).
 * kron: out[c] = a[c / nb] * b[c % nb] (builds wB of the factored loading from the trailing factors). */
size_t cmtfpls_rank1_tensor_workspace_bytes(const int* dims, int n);
int cmtfpls_rank1_tensor_f64(const double* Z, const int* dims, int n, double tol, double* factors, int ld,
                             double* info, int n_squarings, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_kron_f64(const double* a, int na, const double* b, int nb, double* out, void* stream);

/* ---- cross-covariance contraction (not a reference call site: an exact re-association of the loop)
 * xcov: S (M x P, row-major f64) = Y^T X_(0), i.e. S[m, c] = sum_i Y[i*ldy + m] * X[i, c], on the
 * f64 matrix cores (v_mfma_f64_16x16x4_f64), one read of X.  Inside one component u = Y q, hence
 * np.einsum(X, u) = sum_m q_m S[m] (tpls.py:83) and Y.T @ t = S_(0) kron(wA, wB) (tpls.py:100): the
 * inner loop of tpls.py:79-107 can then run on S alone with mode0_contract_f64 / rank1 / score_f64
 * applied to S.  masked != 0: NaN entries of X contribute 0.  Any M (the reference has no limit, tpls.py:100-102): one pass
 * over X per 64 responses, all through the same workspace (sized for min(M, 64) responses).
 * quadform: out[0] = (q - q_old)^T G (q - q_old) = |Y q - Y q_old|^2 for G = Y^T Y (tpls.py:103). */
size_t cmtfpls_xcov_workspace_bytes(int64_t I, int64_t P, int M);
int cmtfpls_xcov_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S,
                     int masked, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_xcov_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S,
                     int masked, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_quadform_f64(const double* G, int M, const double* q, const double* q_old, double* out, void* stream);
/* xcov_deflate (round 3): the deflation X -= t (x) w (tpls.py:109; NaN stays NaN, values rounded to the storage type) AND the
 * cross-covariance S = Y^T X0 of the DEFLATED block (X0: NaN -> 0) AND ssq[0] = |X0|^2, in one read + write of X.  For blocks with
 * missing values inside the cross-covariance loop: their S cannot be carried across a deflation algebraically, so a component
 * cost a read + write (deflation) and a read (rebuild of S); Y is the already deflated Y (with the masked score's row rescale
 * folded into further columns, as for cmtfpls_xcov_*).  S equals deflating first and calling cmtfpls_xcov_* (masked) bit for bit.
 * M <= 64, A * B % 4 == 0 (CMTFPLS_EUNSUPPORTED otherwise).  ws: cmtfpls_xcov_ssq_workspace_bytes(I, A * B, M). */
int cmtfpls_xcov_deflate_f32(float* X, int64_t I, int A, int B, const double* Y, int ldy, int M, const double* t, const double* wA,
                             const double* wB, double* S, double* ssq, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_xcov_deflate_f64(double* X, int64_t I, int A, int B, const double* Y, int ldy, int M, const double* t, const double* wA,
                             const double* wB, double* S, double* ssq, void* ws, size_t ws_bytes, void* stream);
/* xcov_ssq (round 3): S as cmtfpls_xcov_* (unmasked) AND ssq[0] = sum_{i,c} (X[i,c] - mean[c])^2 from the same read of an
 * UNCENTRED X without missing values: |X - X_mean|^2, the denominator of R2X (util.py:7-20, tpls.py:115-117), for the fit that
 * never centres, writes or copies X.  (The f64 value of every element is formed for the matrix cores anyway.) */
size_t cmtfpls_xcov_ssq_workspace_bytes(int64_t I, int64_t P, int M);
/* xcov_stats (round 4): S as cmtfpls_xcov_* (unmasked, M <= 64) AND the statistics pass of tpls.py:61-71 from the same read of an
 * uncentred X: stats[0..P) = the column sums, stats[P..2P) = the column sums of squares (a missing value shows as a NaN in its
 * column's sum: the caller then takes cmtfpls_colstats_* and the masked forms).  With them mean = sum / I and
 * |X - X_mean|^2 = sum_c (sumsq_c - sum_c^2 / I): a fit on the uncentred tensor reads X ONCE before its first component. */
size_t cmtfpls_xcov_stats_workspace_bytes(int64_t I, int64_t P, int M);
int cmtfpls_xcov_stats_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, double* stats,
                           void* ws, size_t ws_bytes, void* stream);
int cmtfpls_xcov_stats_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, double* stats,
                           void* ws, size_t ws_bytes, void* stream);
int cmtfpls_xcov_ssq_f32(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, const double* mean,
                         double* ssq, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_xcov_ssq_f64(const double* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S, const double* mean,
                         double* ssq, void* ws, size_t ws_bytes, void* stream);
/* One inner iteration of the loop on S (tpls.py:80-103 re-associated) issued by a single host call:
 * Z = sum_m q_cur[m] S[m,:] (only if first != 0), rank1(Z) -> (wA, wB, info), q_new = S (wA (x) wB) normalised,
 * du2 = (q_new - q_cur)^T G (q_new - q_cur).  Same results as the separate entries (since round 3 the contraction of the M-row S is
 * one launch without partial rows, and the extraction's last kernel shares a launch with the score: cmtfpls_rank1_score_f64);
 * ws_rank1 as cmtfpls_rank1_workspace_bytes(A, B); ws_contract is no longer used (kept in the signature).  M <= 64. */
int cmtfpls_xcov_iterate_f64(const double* S, int M, int A, int B, const double* q_cur, double* Z, double* wA,
                             double* wB, double* info, int n_squarings, double* q_new, const double* G, double* du2,
                             int first, void* ws_contract, size_t ws_contract_bytes, void* ws_rank1,
                             size_t ws_rank1_bytes, void* stream);
/* The same for SEVERAL coupled blocks and for blocks with missing values (cmtf.py:91-128 re-associated), one host call:
 * per block  Z_b = sum_m q_cur[m] S_b[m,:]  (first != 0 only; masked blocks: Z_b[c] *= n_samples / colcnt[c], missingvals.py:17-19),
 * the loading -- rank1(Z_b) for an order-3 block (A x B), Z_b / |Z_b| for a matrix block (order 2: A = 1, wA = [1] is the caller's) --
 * and tq_b = S2_b (wA (x) wB)  (S2: the cross-covariance with the masked score's row rescale folded into Y; null = S);
 * then q_new = mean_b tq_b (cmtf.py:120; tq of block b is row b of the nb x M matrix `tq`), normalised, and
 * du2 = (q_new - q_cur)^T G (q_new - q_cur).  Blocks of order 2 or 3, M <= 64; ws_rank1: the largest
 * cmtfpls_rank1_workspace_bytes(A, B) of the order-3 blocks. */
typedef struct {
  const double* S;        /* M x A*B */
  const double* S2;       /* M x A*B or null */
  const double* colcnt;   /* A*B observation counts or null (no missing values) */
  double n_samples;       /* rows of X over all ranks (used with colcnt) */
  int order;              /* 2 (matrix block, A == 1) or 3 */
  int A, B;
  int n_squarings;        /* order 3 */
  double* Z;              /* A*B */
  double* wA;             /* A */
  double* wB;             /* B */
  double* info;           /* order 3: {converged, squarings used} */
} cmtfpls_xcov_block;
int cmtfpls_xcov_iterate_blocks_f64(const cmtfpls_xcov_block* blocks, int nb, int M, const double* q_cur, double* tq, double* q_new,
                                    const double* G, double* du2, int first, void* ws_rank1, size_t ws_rank1_bytes, void* stream);
/* S carried across one deflation instead of rebuilt (tpls.py:109 and :113 applied to S = Y^T X_(0)):
 * with X+ = X - t w^T and Y+ = Y - yhat q^T (yhat = T b, the inner-regression prediction),
 *   S+ = S - ya w^T - q v^T,   ya = Y^T t (M, taken before Y is deflated),  v = X+^T yhat (P, from
 * cmtfpls_deflate_contract_yq_* with Y = yhat as an I x 1 matrix and q = [1]),  w[c] = wA[c/B] wB[c%B]. */
int cmtfpls_s_downdate_f64(double* S, int M, int A, int B, const double* ya, const double* wA, const double* wB,
                           const double* q, const double* v, void* stream);
/* v[c] -= sum_{j<k} coef[j] WA[(c/B)*ld + j] WB[(c%B)*ld + j]  (k <= 64; WA: A x ld, WB: B x ld row-major): the xcov loop without
 * writing X.  With X_a = X_0 - sum_{j<a} t_j w_j^T (tpls.py:109 unrolled) the deflated tensor is never formed:
 *   final score   t_a = X_0 w_a - sum_{j<a} t_j (w_j^T w_a)            (cmtfpls_score_* on X_0, then cmtfpls_y_deflate_f64 on t),
 *   down-date     v = X_{a+1}^T yhat = X_0^T yhat - sum_{j<=a} w_j (t_j^T yhat)   (cmtfpls_mode0_contract_* on X_0, then this entry),
 *   R2X           |X_{a+1}|^2 = |X_a|^2 - 2 t^T t_b + t^T t  (t_b: the block's own score, = t for one block):
 * two reads of X per component instead of a read and a read + write, and X_0 stays as centred. */
int cmtfpls_kr_axpy_f64(double* v, int A, int B, const double* WA, const double* WB, int ld, int k, const double* coef, void* stream);
/* axpy_scalar: y[i] -= a[0] * (x ? x[i] : 1) for n doubles, a on the device.  The two rank-one corrections that let the
 * cross-covariance loop run on the caller's UNCENTRED X without ever writing or copying it (round 3):
 *   X_c w = X w - (mean^T w) 1  (scores),   X_c^T yhat = X^T yhat - (1^T yhat) mean  (the down-date of S). */
int cmtfpls_axpy_scalar_f64(double* y, int64_t n, const double* a, const double* x, void* stream);
/* score_contract (round 3): ONE read of X gives a score and the contraction with it,
 *   t[i] = sum_c X[i,c] w[c] - shift[0] - sub_own[i],   c[i] = alpha (t[i] + add_other[i]),   Z = X^T c
 * (w[c] = wA[c/B] wB[c%B]; shift, sub_own, add_other nullable = 0): multi_mode_dot, tpls.py:97-99, followed by
 * np.einsum("i...,i->...", X, c), tpls.py:83, with a score the pass just formed.  It removes the second read of X per component
 * from the never-writing cross-covariance loop above: yhat = T b is a combination of the scores, so X_0^T yhat = sum_j b_j r_j with
 * r_j = X_0^T t_j kept from the pass that formed t_j.  sub_own = T[:, :a] (w_j^T w_a)_j makes t the score of the implicitly
 * deflated X_a; for coupled blocks (cmtf.py:120) add_other = the sum of the other blocks' scores and alpha = 1 / blocks make c the
 * block-averaged score the deflation uses, for the block that is read last.
 * A row lives in the registers of one 1024-thread workgroup (A * B <= 16384) or, beyond that (round 4: the 256 x 256 rows of
 * BASELINE configs[4]), of up to 16 co-resident workgroups that each hold a column slab and exchange the row's partial dot products
 * through 8-byte agent-scope stores (value = flag, no fence): B % (16 / sizeof) == 0, 512 * (16 / sizeof) <= A * B <= 16 * 16384 (f32 with
 * 4096 % B == 0) / 16 * 8192 (f64, other f32 rows), no missing values; CMTFPLS_EUNSUPPORTED otherwise (use cmtfpls_score_* + cmtfpls_mode0_contract_*).
 * csum (nullable): csum[0] = sum_i c[i] (the uncentred form's correction X_c^T c = X^T c - (1^T c) mean).
 * ws: cmtfpls_score_contract_workspace_bytes(I, A * B). */
size_t cmtfpls_score_contract_workspace_bytes(int64_t I, int64_t P);
int cmtfpls_score_contract_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* shift,
                               const double* sub_own, const double* add_other, double alpha, double* t, double* Z, double* csum,
                               void* ws, size_t ws_bytes, void* stream);
int cmtfpls_score_contract_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB, const double* shift,
                               const double* sub_own, const double* add_other, double alpha, double* t, double* Z, double* csum,
                               void* ws, size_t ws_bytes, void* stream);
/* Opt-in mixed-precision forms of xcov and mttkrp for f32-stored X: v_mfma_f32_16x16x4_f32 (half the
 * matrix cycles of the f64 form, HBM-bound instead of matrix-pipe-bound).  X is exact; the other
 * operand is rounded once to f32; f32 accumulation only inside chains of 64 rows (xcov) / 256 columns
 * (mttkrp), every chain added into f64.  Same arguments, workspaces and outputs as the f64 forms. */
int cmtfpls_xcov_f32_mixed(const float* X, int64_t I, int64_t P, const double* Y, int ldy, int M, double* S,
                           int masked, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_mttkrp_f32_mixed(const float* X, int64_t I, int A, int B, const double* WA, const double* WB, int R,
                             double* out, int ldo, void* stream);

/* mttkrp: M (I x R, leading dim ldo) = X_(0) (WA (.) WB), M[i, r] = sum_c X[i,c] WA[c / B, r] WB[c % B, r],
 * WA (A x R) and WB (B x R) row-major f64, R <= 32, on the f64 matrix cores; the Khatri-Rao operand is never
 * materialised (round 3: for A % 16 == 0, B % 64 == 0, R <= 16 the matrix cores contract over one trailing mode with the
 * plain loading of that mode as operand and the other mode's loading is folded into the accumulators; otherwise the
 * product WA[j,r] WB[k,r] is formed from LDS per tile).  One pass over X replaces the R project-and-deflate passes of
 * predict / transform (tpls.py:133-142, 156-165) when X has no NaN:
 * T = M (I + triu(W^T W, 1))^{-1} (the R x R part is done by the caller).
 * CMTFPLS_EUNSUPPORTED when R > 32 or (A + B) * 16 * ceil(R / 16) doubles exceed 152 KB of LDS. */
int cmtfpls_mttkrp_f32(const float* X, int64_t I, int A, int B, const double* WA, const double* WB, int R,
                       double* out, int ldo, void* stream);
int cmtfpls_mttkrp_f64(const double* X, int64_t I, int A, int B, const double* WA, const double* WB, int R,
                       double* out, int ldo, void* stream);

/* ---- K3 score contraction: multi_mode_dot(X, [w...], range(1, X.ndim))  tpls.py:97-99 ---------
 * t[i] = sum_c X[i,c] * wA[c / B] * wB[c % B].
 * rowcnt != NULL selects the masked form miss_mmodedot (missingvals.py:23-38): NaN entries
 * contribute 0 and t[i] = dot / rowcnt[i] * P (0/0 = NaN for an empty row, as the reference). */
int cmtfpls_score_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB,
                      const double* rowcnt, double* t, void* stream);
int cmtfpls_score_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB,
                      const double* rowcnt, double* t, void* stream);
/* score_s: the same contraction for the M rows of a cross-covariance S = Y^T X_(0) inside the xcov loop, tq[m] = S[m, :] . w
 * (= Y^T t, tpls.py:100).  Few long rows (M <= 64, P >= 8192) take one 1024-thread workgroup per row.  Kept apart from
 * cmtfpls_score_*: that kernel sums in another order, and the score of a SAMPLE must not depend on how many samples are
 * passed with it -- M is fixed for a fit, a batch size is not. */
int cmtfpls_score_s_f64(const double* S, int M, int A, int B, const double* wA, const double* wB, double* tq, void* stream);

/* Deflation of one component fused with the first contraction of the next (tpls.py:109 followed by
 * tpls.py:80-83 of the next pass of the component loop): X is deflated in place exactly as
 * cmtfpls_deflate_* does, and in the same sweep Z[c] = sum_i X_new[i,c] * (Y[i,:] . q) and
 * ssq[0] = sum of squares of the observed entries of X_new (R2X numerator) are formed.  One read + one
 * write of X instead of read + write + read.  Vector shapes with B % (16/sizeof(T)) == 0 and M <= 64
 * only (CMTFPLS_EUNSUPPORTED otherwise: deflate, then mode0_contract). */
size_t cmtfpls_deflate_contract_workspace_bytes(int64_t I, int64_t P);
int cmtfpls_deflate_contract_yq_f32(float* X, int64_t I, int A, int B, const double* t, const double* wA,
                                    const double* wB, const double* Y, int ldy, int M, const double* q, double* Z,
                                    int masked, double* ssq, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_deflate_contract_yq_f64(double* X, int64_t I, int A, int B, const double* t, const double* wA,
                                    const double* wB, const double* Y, int ldy, int M, const double* q, double* Z,
                                    int masked, double* ssq, void* ws, size_t ws_bytes, void* stream);
/* score + the partial sums of Y^T t (tpls.py:100, `Y.T @ X_scores`) of every workgroup's rows:
 * qpart is (cmtfpls_sweep_partials() x M) row-major; cmtfpls_q_update_f64 adds the rows in index order.
 * M <= 64 (CMTFPLS_EUNSUPPORTED otherwise: use score + gram_tn). */
int cmtfpls_score_gram_f32(const float* X, int64_t I, int A, int B, const double* wA, const double* wB,
                           const double* rowcnt, double* t, const double* Y, int ldy, int M, double* qpart,
                           void* stream);
int cmtfpls_score_gram_f64(const double* X, int64_t I, int A, int B, const double* wA, const double* wB,
                           const double* rowcnt, double* t, const double* Y, int ldy, int M, double* qpart,
                           void* stream);
/* The Y-side update of one iteration in one launch (tpls.py:100-103), every step optional:
 *   qpart != NULL : q[m] = sum_b qpart[b*M + m]  (b < nblk, fixed order)
 *   normalize != 0: q /= ||q||_2                                            (tpls.py:101)
 *   G != NULL     : du2[0] = (q - q_prev)^T G (q - q_prev) = |Y q - Y q_prev|^2 for G = Y^T Y (tpls.py:103)
 * A sharded fit calls it twice around the all-reduce of q (first the sum, then the rest).  M <= 64. */
int cmtfpls_q_update_f64(const double* qpart, int nblk, int M, double* q, int normalize, const double* G,
                         const double* q_prev, double* du2, void* stream);

/* ---- K6 rank-1 deflation: X -= outer([t, w_J, w_K])  tpls.py:109; cmtf.py:130-131 ------------
 * X[i,c] -= t[i] * wA[c / B] * wB[c % B] in place, one read + one write of X.
 * ssq_part (nullable, cmtfpls_sweep_partials() doubles): per-block partial sums of the squared
 * deflated entries, NaNs skipped: ||X_{a+1}||^2 = the numerator of calcR2X (util.py:13) because
 * X_c - factors_to_tensor(X_factors) IS the deflated tensor at observed positions. */
int cmtfpls_deflate_f32(float* X, int64_t I, int A, int B, const double* t, const double* wA,
                        const double* wB, double* ssq_part, void* stream);
int cmtfpls_deflate_f64(double* X, int64_t I, int A, int B, const double* t, const double* wA,
                        const double* wB, double* ssq_part, void* stream);

/* ---- K3 + K6 fused (transform / predict inner step, tpls.py:133-142, 156-165) -----------------
 * per row: t[i] = score (masked form when rowcnt != NULL), then the row is deflated with it while
 * still in registers.  Returns CMTFPLS_EUNSUPPORTED when a row does not fit one workgroup's
 * registers (P > 65536 for f32, 32768 for f64): call score + deflate instead. */
int cmtfpls_score_deflate_f32(float* X, int64_t I, int A, int B, const double* wA, const double* wB,
                              const double* rowcnt, double* t, double* ssq_part, void* stream);
int cmtfpls_score_deflate_f64(double* X, int64_t I, int A, int B, const double* wA, const double* wB,
                              const double* rowcnt, double* t, double* ssq_part, void* stream);

/* ---- K4 / K5 / K7 / K11 small f64 algebra on tall-skinny operands ----------------------------
 * gram_tn:     C (a x b, row-major) = A^T B over I rows; A is (I x a) with leading dim lda, B is
 *              (I x b) with ldb.  Y.T @ t (tpls.py:100), T^T T and T^T u (normal equations of the
 *              lstsq at tpls.py:110-112), Y^T Y.   Any a, b (tiled 64 x 64).
 * rowdot:      u[i] = sum_m Y[i*ldy + m] * q[m]  (u = Y @ q, tpls.py:102); when u_old != NULL also
 *              du2[0] = sum_i (u_old[i] - u[i])^2   (norm(oldU - u), tpls.py:103).
 * scores_mean: out[i] = (Ts[0][i] + Ts[1][i] + ...) / nb   (np.average(Ts, axis=0), cmtf.py:120).
 * y_deflate:   Y[i,m] -= (sum_r T[i*ldt + r] * b[r]) * q[m]  (tpls.py:113); ssq[0] = ||Y||_F^2 after.
 * sum:         out[0] = sum of n doubles in a fixed order (closes every *ssq_part* array). */
size_t cmtfpls_small_workspace_bytes(void);
int cmtfpls_gram_tn_f64(const double* A, int lda, int a, const double* B, int ldb, int b, int64_t I,
                        double* C, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_rowdot_f64(const double* Y, int ldy, int M, int64_t I, const double* q, double* u,
                       const double* u_old, double* du2, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_scores_mean_f64(const double* Ts, int nb, int64_t I, double* out, void* stream);
int cmtfpls_y_deflate_f64(double* Y, int ldy, int M, int64_t I, const double* T, int ldt, int R,
                          const double* b, const double* q, double* ssq, void* ws, size_t ws_bytes,
                          void* stream);
int cmtfpls_sum_f64(const double* in, int64_t n, double* out, void* stream);

/* project_rows: transform / predict of samples WITH missing values (tpls.py:128-142 with miss_mmodedot, missingvals.py:23-38) from
 * ONE read of the uncentred X, nothing written: a workgroup keeps one row in registers and runs the reference's whole sequence
 * on it -- x - mean (rounded to the storage type, as the centred copy would be), the observation count, and for a = 0..R-1 the
 * masked score t_a = (sum_obs x w_a) * P / n_obs, scores[i*ld + a] = t_a, x -= t_a w_a (rounded to the storage type) --
 * instead of a centring pass and R read + write passes (cmtfpls_score_deflate_*).  WA (A x R), WB (B x R) row-major, mean (P,
 * nullable = 0).  A lane holds up to 16 vectors of 16 bytes: rows of up to 4096 vectors run in 256-thread workgroups, rows of
 * up to 16384 vectors (256 x 256 f32, BASELINE configs[4]) in 1024-thread workgroups of 128 registers per lane.
 * CMTFPLS_EUNSUPPORTED for longer rows, a trailing extent B that does not divide the workgroup stride (256 or 1024 times
 * 16 / sizeof(T) elements), or loadings beyond 144 KB of LDS: the caller keeps the passes.
 * project_rows2: the same for TWO COUPLED blocks sharing the sample mode (ctPLS.transform / predict with missing values,
 * cmtf.py:143-177,180-210): one workgroup holds the sample's row of both blocks, the score of a step is the mean of the two
 * masked block scores (np.average, cmtf.py:155,206) and deflates both.  The shorter block may take at most 4 vectors per lane
 * (an I x 512 matrix block: one), both blocks together at most 17; otherwise CMTFPLS_EUNSUPPORTED. */
int cmtfpls_project_rows_f32(const float* X, int64_t I, int A, int B, int R, const double* WA, const double* WB,
                             const double* mean, double* scores, int ld, void* stream);
int cmtfpls_project_rows_f64(const double* X, int64_t I, int A, int B, int R, const double* WA, const double* WB,
                             const double* mean, double* scores, int ld, void* stream);
int cmtfpls_project_rows2_f32(const float* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                              const float* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                              int64_t I, int R, double* scores, int ld, void* stream);
int cmtfpls_project_rows2_f64(const double* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                              const double* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                              int64_t I, int R, double* scores, int ld, void* stream);
/* project_rows_idx / project_rows2_idx (round 4): the same sequence for the n_rows samples listed in `rows` (device array of
 * sample indices into X and scores) only.  The samples of a batch are independent (tpls.py:128-142 works row by row), so a
 * sample WITHOUT a missing value keeps the score of the one-pass form (cmtfpls_mttkrp_* + cmtfpls_unit_upper_solve_rows_f64)
 * and only the samples with one take the masked sequence: a batch with a few incomplete samples no longer pays the
 * arithmetic-bound sequence for all of them.  Shape limits as above; n_rows = 0 is a no-op. */
int cmtfpls_project_rows_idx_f32(const float* X, const int64_t* rows, int64_t n_rows, int A, int B, int R, const double* WA,
                                 const double* WB, const double* mean, double* scores, int ld, void* stream);
int cmtfpls_project_rows_idx_f64(const double* X, const int64_t* rows, int64_t n_rows, int A, int B, int R, const double* WA,
                                 const double* WB, const double* mean, double* scores, int ld, void* stream);
int cmtfpls_project_rows2_idx_f32(const float* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                                  const float* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                                  const int64_t* rows, int64_t n_rows, int R, double* scores, int ld, void* stream);
int cmtfpls_project_rows2_idx_f64(const double* X0, int A0, int B0, const double* WA0, const double* WB0, const double* mean0,
                                  const double* X1, int A1, int B1, const double* WA1, const double* WB1, const double* mean1,
                                  const int64_t* rows, int64_t n_rows, int R, double* scores, int ld, void* stream);

/* ---- collectives of the sharded loop for callers that drive this C ABI directly -----------------------
 * In-place all-reduce(sum) of a device buffer over the caller's RCCL communicator (an ncclComm_t passed as void*), on
 * `stream`: Z (P doubles) and Y^T t (M doubles) per direct iteration, T^T [T | u] per component, S per component with
 * the cross-covariance form.  The library does not link RCCL: ncclAllReduce is resolved at first use from the RCCL
 * already loaded in the process (the one the communicator came from), else from librccl.so.1 on the loader path;
 * CMTFPLS_EUNSUPPORTED when there is none.  (The Python package issues the same collectives through its own process group.) */
int cmtfpls_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream);
int cmtfpls_allreduce_sum_f32(void* comm, float* buf, size_t count, void* stream);

/* ---- K7 / K8 / projection fix-up without a host round trip ------------------------------------------
 * normal_solve: b (k entries, stride incb) = argmin |T b - u| from the k x k normal equations G b = g with
 *   G = T^T T, g = T^T u (row-major f64, k <= 64): `np.linalg.lstsq(T, u, rcond=-1)[0]` of tpls.py:110-112 /
 *   cmtf.py:135-137 restricted to the k = a + 1 non-zero score columns.  Cholesky of the equilibrated matrix
 *   diag(G)^(-1/2) G diag(G)^(-1/2) (the score columns differ in scale by orders of magnitude; the raw normal
 *   equations would square that spread); a column that is zero or dependent to working precision gets b = 0.
 * unit_upper_solve_rows: rows of M (I x R, leading dim ld) are overwritten by the rows of T solving
 *   T (I + triu(U, 1)) = M - 1 shift^T: the R x R part of the one-pass transform / predict (see cmtfpls_mttkrp_*).
 *   shift (R doubles, nullable = 0): mean^T W, the centring `X - X_mean` of tpls.py:130,153 moved behind the MTTKRP,
 *   (X - 1 mean^T) W = X W - 1 (mean^T W)^T, so that X is read once, uncentred, and never written.  nan_flag (one
 *   int, nullable, zeroed by the caller): set to 1 when M holds a NaN, i.e. a row of X had a missing value.
 * kr_gram: G (R x R) = (first ? 1 : G) .* scale * L^T L for one loading matrix L (n x R row-major): the Gram
 *   matrix of a Khatri-Rao product is the Hadamard product of the mode Grams; call once per mode.
 * khatri_rao: out ((na * nb) x R) = column-wise Kronecker product of Am (na x R) and Bm (nb x R)
 *   (tensorly.tenalg.khatri_rao as used by util.py:19, first matrix varying slowest).
 * recon: Xhat[i, c] = sum_r T[i*ldt + r] * WA[(c / B)*R + r] * WB[(c % B)*R + r] + mean[c]  for I rows, written
 *   in the storage type: factors_to_tensor (util.py:18-20) + X_mean as X_reconstructed uses it (tpls.py:188-189,
 *   cmtf.py:233-237), the Khatri-Rao operand never materialised; mean nullable.  Any shape (16-byte vectors when
 *   B % (16/sizeof(T)) == 0 and `out` is aligned, single elements otherwise). */
int cmtfpls_normal_solve_f64(const double* G, const double* g, int k, double* b, int incb, void* stream);
/* normal_solve_ws: the same solve for any k <= 1024 (the reference's lstsq has no limit on n_components,
 * tpls.py:110-112): k <= 64 is cmtfpls_normal_solve_f64 (ws may be null); beyond, the equilibrated matrix lives in the
 * caller's workspace (cmtfpls_normal_solve_workspace_bytes(k); 0 for k <= 64) -- same algorithm, same pivot rule. */
size_t cmtfpls_normal_solve_workspace_bytes(int k);
int cmtfpls_normal_solve_ws_f64(const double* G, const double* g, int k, double* b, int incb, void* ws, size_t ws_bytes,
                                void* stream);
int cmtfpls_unit_upper_solve_rows_f64(double* M, int64_t I, int ld, int R, const double* U, const double* shift, int* nan_flag,
                                      void* stream);
int cmtfpls_kr_gram_f64(const double* L, int n, int R, double* G, int first, double scale, void* stream);
/* Row a of that Gram matrix only: g[j] = (first ? 1 : g[j]) * sum_i L[i][j] L[i][a] for j < a -- w_j^T w_a, what the never-writing
 * cross-covariance loop needs per component (the score correction T[:, :a] g). */
int cmtfpls_kr_gram_row_f64(const double* L, int n, int R, int a, double* g, int first, void* stream);
int cmtfpls_khatri_rao_f64(const double* Am, int na, const double* Bm, int nb, int R, double* out, void* stream);
/* predict_rows: out[i, m] = mean[m] + sum_a S[i*lds + a] * Bm[a*M + m]: `X_projection @ coef_ @ Q^T + Y_mean` (tpls.py:143,
 * cmtf.py:177) applied to the device-resident scores, Bm = coef_ Q^T (R x M, formed by the caller); mean nullable. */
int cmtfpls_predict_rows_f64(const double* S, int64_t I, int lds, int R, const double* Bm, int M, const double* mean,
                             double* out, int ldo, void* stream);
int cmtfpls_recon_f32(const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                      const double* mean, float* out, void* stream);
int cmtfpls_recon_f64(const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB, int A, int B,
                      const double* mean, double* out, void* stream);
/* recon_r2: the two sums of calcR2X(X - mean, factors_to_tensor(X_factors)) (util.py:7-15 as called at tpls.py:115-117,
 * cmtf.py:132-134) in ONE read of the original X, the reconstruction never materialised:
 *   out[0] = sum over finite x of (xhat - x)^2,  out[1] = sum over finite x of x^2,  x = X[i,c] - mean[c] (mean
 * nullable), xhat as in recon;  R2X = 1 - out[0] / out[1].  (The fit itself gets R2X from the deflation sweep; this
 * is the literal formula for callers of calcR2X and for checking that identity at full size.)  Any shape; R <= 16. */
size_t cmtfpls_recon_r2_workspace_bytes(int64_t I, int64_t P);
int cmtfpls_recon_r2_f32(const float* X, const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB,
                         int A, int B, const double* mean, double* out, void* ws, size_t ws_bytes, void* stream);
int cmtfpls_recon_r2_f64(const double* X, const double* T, int64_t I, int ldt, int R, const double* WA, const double* WB,
                         int A, int B, const double* mean, double* out, void* ws, size_t ws_bytes, void* stream);

/* ---- leave-one-out refits, all folds in one launch: validate.get_q2y  (cmtf_pls/validate.py:7-37) ------------
 * For every fold i in [fold0, fold0 + nfolds): a complete tPLS fit (tpls.py:73-113; R components, tol, max_iter, the
 * reference's loop and convergence test) on the I - 1 samples other than i, then predict (tpls.py:122-143) of sample
 * i into Ypred[i, :].  One workgroup per fold; X (I x A*B) and Y (I x M) are the ORIGINAL float64 data; colsum_x /
 * colsum_y are their column sums (fold means are down-dated from them).  n_iter (nullable, I x R ints): inner
 * iterations executed.  ws >= nfolds * cmtfpls_loo_fold_workspace_bytes(...).  X of order 2 (A = 1) or 3 without
 * missing values; CMTFPLS_EUNSUPPORTED when min(A, B) > 64, M > 64, R > 16 or the per-fold vectors exceed the LDS
 * (the caller then refits per fold with the regular entry points). */
size_t cmtfpls_loo_fold_workspace_bytes(int I, int A, int B, int M, int R);
int cmtfpls_loo_tpls_f64(const double* X, const double* Y, const double* colsum_x, const double* colsum_y, int I, int A,
                         int B, int M, int R, double tol, int max_iter, int fold0, int nfolds, double* Ypred,
                         int* n_iter, void* ws, size_t ws_bytes, void* stream);
/* loo_xcov (round 4): the same leave-one-out refits for trailing shapes BEYOND the LDS-resident form -- min(A, B) <= 256 (128 x 128,
 * 256 x 256), M <= 128 (second session of round 4: 64 before), R <= 64 -- one 1024-thread workgroup per fold.  The fold's means are down-dated from the column sums; inside a
 * component the NIPALS loop runs on the fold's cross-covariance S = Y_f^T X_f (M x P, formed once per component: tpls.py:83 becomes
 * S^T q, tpls.py:100 becomes S w, tpls.py:103 the quadratic form with Y_f^T Y_f), so an inner iteration reads 2 M P doubles
 * instead of the fold's 2 I P; the rank-1 extraction squares the n x n Gram matrix on the f64 matrix cores inside the workgroup.
 * Same arguments and results as cmtfpls_loo_tpls_f64 (which it equals to rounding where both apply);
 * ws >= nfolds * cmtfpls_loo_xcov_fold_workspace_bytes(...) (a centred copy of X per resident fold, deflated in place). */
size_t cmtfpls_loo_xcov_fold_workspace_bytes(int I, int A, int B, int M, int R);
int cmtfpls_loo_xcov_f64(const double* X, const double* Y, const double* colsum_x, const double* colsum_y, int I, int A,
                         int B, int M, int R, double tol, int max_iter, int fold0, int nfolds, double* Ypred,
                         int* n_iter, void* ws, size_t ws_bytes, void* stream);
/* fit_small: the COMPLETE tPLS.fit (tpls.py:73-120: preprocess, every component's NIPALS loop with its convergence test,
 * rank-1 extraction, deflation, inner regression, Y deflation) of a small problem in ONE launch of one workgroup -- a fit of
 * BASELINE configs[0] (200 x 10 x 8, R = 3) is otherwise a few hundred launches of pure latency.  float64, X of order 2 or 3
 * (A = 1 for a matrix) WITHOUT missing values; limits as loo_tpls: min(A, B) <= 64, M <= 64, R <= 16, the workgroup's vectors
 * within 150 KB of LDS (CMTFPLS_EUNSUPPORTED otherwise).  Outputs (device): T (I x R), U (I x R), WA (A x R), WB (B x R),
 * Q (M x R), coef (R x R), ssq ((R + 1) x 2: row 0 = |X_c|^2, |Y_c|^2, row a + 1 = the deflated norms after component a, so
 * R2X[a] = 1 - ssq[a+1][0] / ssq[0][0] and R2Y likewise, tpls.py:115-120), x_mean (A * B), y_mean (M), n_iter (R ints),
 * flag (one int, zeroed by the caller: set to 1, nothing else written, when X or Y holds a non-finite value).
 * ws: cmtfpls_fit_small_workspace_bytes (the centred working copies of X and Y). */
size_t cmtfpls_fit_small_workspace_bytes(int I, int A, int B, int M);
int cmtfpls_fit_small_f64(const double* X, const double* Y, int I, int A, int B, int M, int R, double tol, int max_iter,
                          double* T, double* U, double* WA, double* WB, double* Q, double* coef, double* ssq, double* x_mean,
                          double* y_mean, int* n_iter, int* flag, void* ws, size_t ws_bytes, void* stream);

/* ---- synthetic inputs on the device: cmtf_pls/synthetic.py:59-74 (import_synthetic), :5-34 (make_synthetic_test)
 * The dense CP tensor of the drawn factors is cmtfpls_recon_* with T = the sample factor; add_noise then adds
 * sigma * N(0,1) in place (synthetic.py:71,74) and, if nan_fraction > 0, plants an i.i.d. NaN mask (BASELINE
 * configs[3]).  The generator is counter based (Philox4x32-10, key = seed, counter = (offset + e) / 4 for
 * element e of this buffer; offset = the buffer's first GLOBAL element index): a row shard of a
 * tensor receives exactly the noise those rows have in the whole tensor. */
int cmtfpls_add_noise_f32(float* X, int64_t n, double sigma, uint64_t seed, uint64_t offset, double nan_fraction,
                          void* stream);
int cmtfpls_add_noise_f64(double* X, int64_t n, double sigma, uint64_t seed, uint64_t offset, double nan_fraction,
                          void* stream);

/* ---- measured HBM ceilings (the roofline denominator "re-measured with a device copy
 * kernel"; not a reference call site) ---------------------------------------------------------------
 * Plain 16-byte-per-lane non-temporal streaming kernels over `bytes` of a 16-byte-aligned buffer:
 * read (sum kept in sink[0..blocks)), in-place read-modify-write (negates: two calls restore the data),
 * copy src -> dst, each under one of four workgroup -> address maps:
 *   0 flat (consecutive workgroups adjacent, grid stride; row_bytes ignored),
 *   1 chunked (a 256-thread workgroup owns row_bytes contiguous bytes at a time, grid stride over rows),
 *   2 row per 1024-thread workgroup, barrier between the read burst and the write burst (row_bytes = 16 KB * {1,2,4,8}):
 *     the map of deflate_rows / center_rows / score_deflate,
 *   3 column owner: a 256-thread workgroup owns 8 KB of columns and a block of rows, 4 rows in flight (row_bytes a
 *     multiple of 8 KB; `blocks` = total workgroups): the map of the contraction and of deflate_contract.
 * blocks <= cmtfpls_ceiling_max_blocks(). */
int cmtfpls_ceiling_max_blocks(void);
int cmtfpls_ceiling_read(const void* buf, size_t bytes, int64_t row_bytes, int map, float* sink, int blocks, void* stream);
int cmtfpls_ceiling_rmw(void* buf, size_t bytes, int64_t row_bytes, int map, int blocks, void* stream);
int cmtfpls_ceiling_copy(const void* src, void* dst, size_t bytes, int64_t row_bytes, int map, int blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMTFPLS_H */
