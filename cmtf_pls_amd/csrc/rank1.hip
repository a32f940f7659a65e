// Rank-1 extraction of the cross-covariance matrix Z (A x B, f64): the leading singular pair,
// i.e. what parafac(Z, 1, init="svd", normalize_factors=True) returns for a matrix
// (tpls.py:86-88, cmtf.py:100-102).  LAPACK is not available on the device and a plain power
// iteration converges at (s2/s1)^2 per step, so the Gram matrix G of the smaller side is squared
// repeatedly instead: after s squarings G^(2^s) is rank one to (s2/s1)^(2^(s+1)); each squaring is
// one n x n x n f64 product (n <= 256 for the benchmark shapes) spread over (n/16)^2 workgroups, each
// 16 x 16 tile on the f64 matrix cores with its operands loaded straight into the MFMA layout.  Scaling between
// squarings is by an exact power of two taken from the trace, so the iteration is
// bit-reproducible; it stops early (later launches return at once) when tr(G^2) == tr(G)^2 to
// 1e-13.  The dominant column of the final G then seeds one exact pass y = M^T seed, x = M y (= G_0 seed) with
// Z itself (M = Z or Z^T), which also gives exact zeros in the loadings wherever Z has an all-zero
// row or column (tests/test_tpls.py:98-104 relies on that).
#include "common.hpp"
#include <atomic>

namespace cmtfpls {

constexpr int kTile = 16;
constexpr int kMaxTiles = 256;  // n <= 4096 (round 3: the control block is sized by the call, not by this bound)
constexpr int kMaxSteps = 48;

// Control block at the head of the workspace: four flags, then trace[(kMaxSteps + 2) x nt] (trace[s * nt + tile]: the
// diagonal-tile partial traces of G_s) and fro[2 x nt^2] (fro[(s & 1) * nt^2 + tile]: per-tile sums of squares of G_s,
// |G_s|_F^2 = tr(G_s^2)), nt = ceil(n / 16).
struct Rank1Ctl {
  int done;                                // set once G is numerically rank one
  int final_buf;                           // which ping-pong buffer holds the final G
  int steps_used;                          // squarings actually computed
  int last_step;                           // >= 0: the output of this step is the final G (set by that step itself)
  double pad[6];                           // the arrays start 64 bytes in
};
__host__ __device__ __forceinline__ size_t rank1_ctl_bytes(int nt) {
  return sizeof(Rank1Ctl) + ((size_t)(kMaxSteps + 2) * nt + 2 * (size_t)nt * nt) * sizeof(double);
}
__device__ __forceinline__ double* ctl_trace(Rank1Ctl* c, int nt, int step) { return reinterpret_cast<double*>(c + 1) + (size_t)step * nt; }
__device__ __forceinline__ double* ctl_fro(Rank1Ctl* c, int nt, int which) {
  return reinterpret_cast<double*>(c + 1) + (size_t)(kMaxSteps + 2) * nt + (size_t)which * nt * nt;
}

__device__ __forceinline__ double pow2_scale_from_trace(const double* parts, int nt, double* tr_out) {
  double tr = 0.0;
  for (int i = 0; i < nt; ++i) tr += parts[i];
  *tr_out = tr;
  if (!(tr > 0.0) || !isfinite(tr)) return 1.0;
  int e;
  frexp(tr, &e);            // tr = m * 2^e, m in [0.5, 1)
  return ldexp(1.0, -e);    // exact power of two
}

typedef double d4r_t __attribute__((ext_vector_type(4)));

// C = s^2 * M M^T for row-major M (n x k, leading dim ld); s is the power-of-two scale derived from
// the trace of the input (step >= 1) or 1 (step 0, the Gram matrix of Z itself).
//
// One workgroup (4 wavefronts) per 16 x 16 tile of C on the f64 matrix cores, operands straight from
// global memory into registers in the MFMA layout (no LDS staging):
//   v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][kq = l >> 4] and B[kq][j = l & 15] and holds
//   D[(l >> 4) + 4 e][l & 15], e = 0..3.
//   A k-chunk is 128 columns; wavefront w takes columns [32 w, 32 w + 32) of it, lane group kq the 8
//   consecutive columns 32 w + 8 kq + (0..7): MFMA s of the chunk multiplies column 32 w + 8 kq + s of
//   row i0 + (l & 15) with the same column of row j0 + (l & 15) (B = M^T: B[kq][j] = M[j0 + j][column]).
//   The 4 wavefronts' partial tiles are added in wavefront order through LDS: the summation order is fixed
//   (bit-reproducible), and tile (i, j) is bitwise the transpose of tile (j, i) (same products, same order).
// A step is a chain of memory latencies (the previous step's output lives in another XCD's L2), so the
// kernel issues the 16 operand loads of its first chunk first and only then reads the control block (traces,
// Frobenius partials) that decides the scale and the early exit: one latency per step instead of three.
//
// Convergence (step >= 1, from the INPUT G_{s-1}): rho = tr(G^2) / tr(G)^2 = 1 - 2 (lambda_2/lambda_1) to first
// order.  rho >= 1 - 1e-13: the input is rank one to 5e-14 and is taken as the result (this launch and the
// later ones return at once).  rho >= 1 - 1e-7: the input's ratio is <= 5e-8, so THIS step's output has ratio
// <= 2.5e-15: the product is computed and declared final (ctl->done), which saves the launch that would only
// have detected it.  In both cases the dominant column still goes through one exact pass with Z afterwards.
__global__ __launch_bounds__(kTile* kTile) void syrk_step_kernel(const double* __restrict__ M, int n, int k, int ld,
                                                                double* __restrict__ C, Rank1Ctl* __restrict__ ctl,
                                                                int step, int out_buf, double* __restrict__ C_keep) {
  __shared__ double red[4][kTile * kTile];
  __shared__ double diag[kTile];
  __shared__ double s_scale;
  __shared__ int s_done;
  __shared__ double fsum[4];
  __shared__ double tr_s[kMaxTiles];
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kTile + tx;
  const int lane = tid & 63, wv = tid >> 6, ri = lane & 15, kq = lane >> 4;
  const int nt = (n + kTile - 1) / kTile;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;

  // (1) operands of the first k-chunk: unconditional loads (clamped address, select afterwards) so that
  // all 16 are in flight together
  constexpr int KC = 128, KW = 8;                     // chunk width; columns per lane per chunk
  const bool ra = (i0 + ri) < n, rb = (j0 + ri) < n;
  const double* __restrict__ rowa = M + (int64_t)(ra ? i0 + ri : 0) * ld;
  const double* __restrict__ rowb = M + (int64_t)(rb ? j0 + ri : 0) * ld;
  const int cl = 32 * wv + KW * kq;                   // this lane's first column inside a chunk
  double a[KW], b[KW];
#pragma unroll
  for (int s2 = 0; s2 < KW; ++s2) {
    const int c = cl + s2;
    const int cc = (c < k) ? c : 0;
    a[s2] = rowa[cc];
    b[s2] = rowb[cc];
  }

  // (2) control: scale from the input's trace, early exit once the input is numerically rank one.
  // The control words are vector loads issued right behind the operand loads (same single wait).
  double fro_in = 0.0;
  if (step >= 1) {
    // the flag's address is laundered into a VGPR so that this is a vector load in the same batch as
    // the others (as a scalar load the compiler issues it after the wait: a second latency)
    int zero = 0;
    asm volatile("" : "+v"(zero));
    // done: an earlier launch found its input rank one.  last_step = s0 < step: step s0 declared its own output
    // final.  (A workgroup of step s0 itself may see last_step == s0, written by a sibling: it still computes.)
    const int last_in = (&ctl->last_step)[zero];
    const int done_in = (&ctl->done)[zero] | ((last_in >= 0 && last_in < step) ? 1 : 0);
    const double* fp = ctl_fro(ctl, nt, (step - 1) & 1);
    const double trv = ctl_trace(ctl, nt, step - 1)[(tid < nt) ? tid : 0];   // nt <= 256 = the workgroup size
    // |G_{s-1}|_F^2 from the per-tile sums the previous step left (fixed order: bit-reproducible)
    double f = fp[(tid < nt * nt) ? tid : 0];
    if (tid < nt) tr_s[tid] = trv;
    if (tid >= nt * nt) f = 0.0;
    for (int i = tid + kTile * kTile; i < nt * nt; i += kTile * kTile) f += fp[i];
    f = wave_sum(f);
    if ((tid & 63) == 0) fsum[tid >> 6] = f;
    if (tid == 0) s_done = done_in;
    __syncthreads();
    fro_in = ((fsum[0] + fsum[1]) + fsum[2]) + fsum[3];
  }
  if (tid == 0) {
    int done = 0;
    double scale = 1.0;
    if (step == 0) {
      if (blockIdx.x == 0 && blockIdx.y == 0) { ctl->done = 0; ctl->final_buf = -1; ctl->steps_used = 0; ctl->last_step = -1; }
    } else {
      done = s_done;
      if (!done) {
        // G_s = (sc_{s-1} G_{s-1})^2 with sc_{s-1} the power-of-two scale of tr(G_{s-1}).
        double tr1;
        scale = pow2_scale_from_trace(tr_s, nt, &tr1);
        const double rho = fro_in / (tr1 * tr1);
        const bool rank_one_in = !(tr1 > 0.0) || rho >= 1.0 - 1e-13;     // the input already is the result
        const bool rank_one_out = rho >= 1.0 - 1e-7;                      // this step's output will be
        if (blockIdx.x == 0 && blockIdx.y == 0) {
          if (rank_one_in) { ctl->done = 1; ctl->final_buf = out_buf ^ 1; }
          else {
            ctl->steps_used = step;
            if (rank_one_out) { ctl->last_step = step; ctl->final_buf = out_buf; }
          }
        }
        if (rank_one_in) done = 1;
      }
    }
    s_scale = scale;
    s_done = done;
  }
  __syncthreads();
  if (s_done) return;
  const double scale = s_scale;

  // (3) the product, chunk by chunk (one chunk whenever k <= 128: every squaring of a 128 x 128 Z)
  d4r_t acc = d4r_t{0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < k; kk += KC) {
    if (kk > 0) {
#pragma unroll
      for (int s2 = 0; s2 < KW; ++s2) {
        const int c = kk + cl + s2;
        const int cc = (c < k) ? c : 0;
        a[s2] = rowa[cc];
        b[s2] = rowb[cc];
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < KW; ++s2) {
      const bool cok = (kk + cl + s2) < k;
      const double av = (ra && cok) ? a[s2] : 0.0;
      const double bv = (rb && cok) ? b[s2] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][(kq + 4 * e) * kTile + ri] = acc[e];
  __syncthreads();
  // thread (ty, tx) closes element (row ty, column tx) of the tile
  double cacc = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  cacc *= scale * scale;
  const bool inside = (i0 + ty < n && j0 + tx < n);
  if (inside) C[(int64_t)(i0 + ty) * n + (j0 + tx)] = cacc;
  if (inside && C_keep) C_keep[(int64_t)(i0 + ty) * n + (j0 + tx)] = cacc;   // step 0: G_0 = M M^T is kept for the finish
  {
    // this tile's contribution to |G_s|_F^2
    double sq = inside ? cacc * cacc : 0.0;
    sq = wave_sum(sq);
    if ((tid & 63) == 0) fsum[tid >> 6] = sq;   // fsum is free: its readers passed two barriers since
    __syncthreads();
    if (tid == 0) ctl_fro(ctl, nt, step & 1)[blockIdx.y * nt + blockIdx.x] = ((fsum[0] + fsum[1]) + fsum[2]) + fsum[3];
  }
  if (blockIdx.x == blockIdx.y) {
    if (tx == ty) diag[tx] = (i0 + ty < n) ? cacc : 0.0;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int l = 0; l < kTile; ++l) t += diag[l];
      ctl_trace(ctl, nt, step)[blockIdx.x] = t;
    }
  }
}

// ---- the whole chain of squarings in ONE launch (round 4) -------------------------------------------------------------------
// A launch per squaring costs a dispatch (2.5 us back to back) plus its own ramp: 6.4 us per step inside a fit for 0.5 us of
// matrix work (profiles/r04o_*).  Here the (n/16)^2 workgroups of the step kernel stay resident (n <= 256: at most 256 workgroups
// of 256 threads) and walk steps 0 .. n_squarings themselves.  What a step needs from the other workgroups -- two 16-row panels
// of G_{s-1}, the diagonal tiles' traces, every tile's sum of squares -- is passed WITHOUT a grid barrier and without release
// fences: each G_s has its own buffer, preset to an all-ones pattern no product yields (one preset launch per extraction), every element
// is written with an 8-byte agent-scope store and IS its own flag; a consumer simply loads its MFMA operands with agent-scope
// loads (which by-pass the XCD's L2: measured one-way 0.4 us, tools/exp/xch_latency.hip) and repeats the batch while a lane still
// sees the preset.  Same tile arithmetic, same summation orders, same control decisions as syrk_step_kernel -- every workgroup
// derives them from the same bits -- so the chain is bit-identical to the launch-per-step form.  Spins are bounded: a wavefront
// that gives up continues on NaNs (which it publishes: nobody else waits), and the extraction reports "not converged".
constexpr unsigned long long kChainEmpty = ~0ull;
constexpr int kChainMaxSteps = 31;            // squarings of one chain launch (the engine's budget ceiling is 30)
constexpr int kChainSpin = 1 << 15;           // tries before a wavefront gives up (tens of milliseconds; a healthy wait is a few microseconds)

__device__ __forceinline__ double chain_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void chain_st(double* p, double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (b == kChainEmpty) b = 0x7FF8000000000000ull;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool chain_empty(double v) { return (unsigned long long)__double_as_longlong(v) == kChainEmpty; }

// the preset as an ordinary kernel (a kernel node under graph capture like every other launch of the sequence)
__global__ __launch_bounds__(256) void chain_preset_kernel(unsigned long long* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = kChainEmpty;
}

// gbufs: (n_squarings + 1) matrices of n x n (G_s at s * n * n); trs: (n_squarings + 1) x nt; fros: (n_squarings + 1) x nt * nt --
// one contiguous region preset to 0xFF bytes by the caller.
#ifdef CMTFPLS_CHAIN_PROFILE       // tools/exp builds only: workgroup (0, 0) stamps the phases of every step (100 MHz clock)
__device__ unsigned long long g_chain_prof[1024];
#define CHAIN_STAMP(slot) do { if (first_wg && tid == 0 && step < 32) g_chain_prof[step * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CHAIN_COUNT(slot, v) do { if (first_wg && tid == 0 && step < 32) g_chain_prof[step * 8 + (slot)] = (unsigned long long)(v); } while (0)
#else
#define CHAIN_STAMP(slot) do { } while (0)
#define CHAIN_COUNT(slot, v) do { } while (0)
#endif

__global__ __launch_bounds__(kTile* kTile) void syrk_chain_kernel(const double* __restrict__ M0, int n, int k0,
                                                                 double* gbufs, double* trs, double* fros, double* gave_up,
                                                                 Rank1Ctl* __restrict__ ctl, int n_squarings) {
  constexpr int KC = 128, KW = 8, LDP = KC + 2;            // chunk width; columns per lane per chunk; padded panel row in LDS
  __shared__ __attribute__((aligned(16))) double pan[2][kTile][LDP];   // the two 16-row panels of the current chunk (steps >= 1)
  __shared__ double red[2][4][kTile * kTile];              // (by step parity: no barrier between a step's last read and the next step's writes)
  __shared__ double diag[2][kTile];
  __shared__ double fsum[2][4];
  __shared__ double cfro[2][4];                            // control words of the step's input: per-wavefront sums of the tiles' sums of squares,
  __shared__ double ctr[2][kTile];                         // the diagonal tiles' traces,
  __shared__ int cdead[2];                                 // and whether a wavefront of this workgroup gave up
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kTile + tx;
  const int lane = tid & 63, wv = tid >> 6, ri = lane & 15, kq = lane >> 4;
  const int nt = (n + kTile - 1) / kTile, nt2 = nt * nt;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  const bool first_wg = blockIdx.x == 0 && blockIdx.y == 0;
  const bool ra = (i0 + ri) < n, rb = (j0 + ri) < n;
  const int cl = 32 * wv + KW * kq;
  const int64_t nn = (int64_t)n * n;
  const double qnan = __longlong_as_double(0x7FF8000000000000ll);
  bool dead = false;                                       // this wavefront gave up on a partner
  if (first_wg && tid == 0) { ctl->done = 0; ctl->final_buf = -1; ctl->steps_used = 0; ctl->last_step = -1; }
  if (tid < 2) cdead[tid] = 0;
  // ---- step 0: G_0 = M0 M0^T, operands straight from global memory (Z was written before this launch) ----
  // ---- step s >= 1: the panels of G_{s-1} are fetched by the WHOLE workgroup in coalesced 8-byte agent-scope loads (element
  // e = tid + 256 i of a 16 x 128 chunk: 512 contiguous bytes per instruction -- as MFMA operands straight from memory every
  // load instruction would touch 64 lines, and un-cached loads are paid per line), each wavefront repeating its own batch
  // while one value still shows the preset; then staged through LDS into the MFMA layout.  The control words ride in the same
  // batch: wavefront w takes the tiles' sums of squares 64 w .. 64 w + 63, wavefront 0 the traces; after the staging barrier
  // EVERY thread derives the scale and the exit decisions from the same LDS words, in syrk_step_kernel's orders -- fro = ((w0 +
  // w1) + w2) + w3 with w_i the butterfly sum of entries 64 i .. 64 i + 63, the trace added in tile order -- while the operands
  // are on their way from LDS to the matrix cores (one wavefront deciding for all cost 0.6 us of every step).
  int last_step = -1;                                      // (what syrk_step_kernel keeps in ctl->last_step)
  for (int step = 0; step <= n_squarings; ++step) {
    if (last_step >= 0) break;                             // the previous step declared its own output final
    const int par = step & 1;
    const double* __restrict__ M = step == 0 ? M0 : gbufs + (int64_t)(step - 1) * nn;
    const int k = step == 0 ? k0 : n, ld = k;
    double* C = gbufs + (int64_t)step * nn;
    double a[KW], b[KW];
    double scale = 1.0;
    bool input_is_result = false;
    d4r_t acc = d4r_t{0.0, 0.0, 0.0, 0.0};
    CHAIN_STAMP(0);
    for (int kk = 0; kk < k; kk += KC) {
      if (step == 0) {
        const double* __restrict__ rowa = M + (int64_t)(ra ? i0 + ri : 0) * ld;
        const double* __restrict__ rowb = M + (int64_t)(rb ? j0 + ri : 0) * ld;
#pragma unroll
        for (int s2 = 0; s2 < KW; ++s2) {
          const int c = kk + cl + s2;
          const int cc = (c < k) ? c : 0;
          a[s2] = rowa[cc];
          b[s2] = rowb[cc];
        }
      } else {
        double va[8], vb[8], trv = 0.0, f = 0.0;
        const bool first_chunk = kk == 0;
        const double* tp = trs + (int64_t)(step - 1) * nt;
        const double* fp = fros + (int64_t)(step - 1) * nt2;
        unsigned offa[8], offb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = tid + 256 * i, r = e >> 7, c = kk + (e & 127);
          const int cc = (c < k) ? c : 0;
          const int rra = (i0 + r < n) ? i0 + r : 0, rrb = (j0 + r < n) ? j0 + r : 0;
          offa[i] = (unsigned)(rra * ld + cc);
          offb[i] = (unsigned)(rrb * ld + cc);
        }
        int spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 8; ++i) {                     // (uniform base + 32-bit lane offset: n <= 256)
            va[i] = chain_ld(M + offa[i]);
            vb[i] = chain_ld(M + offb[i]);
          }
          if (first_chunk) {
            f = chain_ld(fp + ((tid < nt2) ? tid : 0));
            ok = !chain_empty(f);
            if (wv == 0) {
              trv = chain_ld(tp + ((lane < nt) ? lane : 0));
              ok = ok && !chain_empty(trv);
            }
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) ok = ok && !chain_empty(va[i]) && !chain_empty(vb[i]);
          if (dead || __all(ok)) break;
          if (++spins > kChainSpin) {                      // a partner is not resident (the GPU is shared): say so, go on with NaNs
            dead = true;
            if (lane == 0) chain_st(gave_up, 1.0);
            break;
          }
        }
        CHAIN_STAMP(1);
        CHAIN_COUNT(7, spins);
        if (kk > 0) __syncthreads();                       // the previous chunk's panels have been consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = tid + 256 * i, r = e >> 7, c = e & 127;
          pan[0][r][c] = dead ? qnan : va[i];
          pan[1][r][c] = dead ? qnan : vb[i];
        }
        if (first_chunk) {
          const double fw = wave_sum((tid < nt2) ? f : 0.0);
          if (lane == 0) cfro[par][wv] = fw;
          if (wv == 0 && lane < nt) ctr[par][lane] = trv;
          if (dead && lane == 0) cdead[par] = 1;
        }
        __syncthreads();
        CHAIN_STAMP(2);
        if (first_chunk) {
          // (a workgroup with a wavefront that gave up decides "go on": the chain then walks to its last step on NaNs)
          const double fro_in = ((cfro[par][0] + cfro[par][1]) + cfro[par][2]) + cfro[par][3];
          double tr1 = 0.0;
          for (int i = 0; i < nt; ++i) tr1 += ctr[par][i];
          if ((tr1 > 0.0) && isfinite(tr1)) {
            int e;
            frexp(tr1, &e);
            scale = ldexp(1.0, -e);
          }
          const bool gone = cdead[par] != 0;
          const double rho = fro_in / (tr1 * tr1);
          const bool rank_one_in = !gone && (!(tr1 > 0.0) || rho >= 1.0 - 1e-13);     // the input already is the result
          const bool rank_one_out = !gone && rho >= 1.0 - 1e-7;                        // this step's output will be
          if (first_wg && tid == 0) {
            if (rank_one_in) { ctl->done = 1; ctl->final_buf = step - 1; }
            else {
              ctl->steps_used = step;
              if (rank_one_out) { ctl->last_step = step; ctl->final_buf = step; }
            }
          }
          if (rank_one_in) { input_is_result = true; break; }
          if (rank_one_out) last_step = step;
        }
#pragma unroll
        for (int s2 = 0; s2 < KW; ++s2) {
          a[s2] = pan[0][ri][cl + s2];
          b[s2] = pan[1][ri][cl + s2];
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < KW; ++s2) {
        const bool cok = (kk + cl + s2) < k;
        const double av = (ra && cok) ? a[s2] : 0.0;
        const double bv = (rb && cok) ? b[s2] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
    }
    if (input_is_result) break;
    CHAIN_STAMP(3);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[par][wv][(kq + 4 * e) * kTile + ri] = acc[e];
    __syncthreads();
    CHAIN_STAMP(4);
    double cacc = ((red[par][0][tid] + red[par][1][tid]) + red[par][2][tid]) + red[par][3][tid];
    cacc *= scale * scale;
    const bool inside = (i0 + ty < n && j0 + tx < n);
    // (the tile goes out first: moving its stores behind the barrier below -- which waits for them -- made a step slower, and so
    // did limiting the registers for three workgroups per CU: 54 / 57 / 63 us per extraction, profiles/r04w_chain_variants.txt)
    if (inside) chain_st(C + (int64_t)(i0 + ty) * n + (j0 + tx), cacc);
    CHAIN_STAMP(5);
    double sq = inside ? cacc * cacc : 0.0;
    sq = wave_sum(sq);
    if ((tid & 63) == 0) fsum[par][tid >> 6] = sq;
    if (blockIdx.x == blockIdx.y && tx == ty) diag[par][tx] = (i0 + ty < n) ? cacc : 0.0;
    __syncthreads();
    if (tid == 0) {
      chain_st(fros + (int64_t)step * nt2 + blockIdx.y * nt + blockIdx.x, ((fsum[par][0] + fsum[par][1]) + fsum[par][2]) + fsum[par][3]);
      if (blockIdx.x == blockIdx.y) {
        double t = 0.0;
        for (int l = 0; l < kTile; ++l) t += diag[par][l];
        chain_st(trs + (int64_t)step * nt + blockIdx.x, t);
      }
    }
    CHAIN_STAMP(6);
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ Z, int A, int B, double* __restrict__ Zt) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)A * B) return;
  const int b = (int)(idx / A), a = (int)(idx % A);   // Zt is B x A
  Zt[idx] = Z[(int64_t)a * B + b];
}

// ---- finish: seed = dominant column of G (normalised); y = M^T seed; x = M y = G_0 seed ---------
// F1: every workgroup derives the seed redundantly (n <= 1024: cheap, deterministic).  The first ny_blocks
//     workgroups then produce 32 entries of y = M^T seed each (8 row groups per column); the others produce 4
//     entries of x each.  x = M (M^T seed) = G_0 seed with G_0 = M M^T kept from step 0, so x does not wait for y:
//     one launch instead of two (round 2).  A zero row of M is a zero row of G_0: its entry of x is exactly 0, as
//     a zero column of M gives an exact 0 in y (tests/test_tpls.py:98-104 relies on both).
__global__ __launch_bounds__(256) void rank1_seed_xy_kernel(const double* __restrict__ M, int n, int k,
                                                           const double* __restrict__ gbase, int64_t gstride,
                                                           const double* __restrict__ G0,
                                                           const Rank1Ctl* __restrict__ ctl, int last_buf, int ny_blocks,
                                                           double* __restrict__ y, double* __restrict__ x) {
  extern __shared__ double seed[];        // n doubles
  __shared__ double red[16];
  __shared__ double rsum[8][33];
  __shared__ double bestv[4];
  __shared__ int besti[4];
  const int fb = (ctl->final_buf >= 0) ? ctl->final_buf : last_buf;     // ping-pong buffer (launch per step) or step index (chain)
  const double* G = gbase + (int64_t)fb * gstride;
  // argmax of the diagonal (first index on ties)
  double bv = -1.0;
  int bi = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double d = G[(int64_t)i * n + i];
    if (d > bv) { bv = d; bi = i; }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ov = __shfl_xor(bv, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bestv[threadIdx.x >> 6] = bv; besti[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = bestv[0];
  bi = besti[0];
  for (int w = 1; w < 4; ++w)
    if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
  // seed = row bi of G (G is bitwise symmetric), normalised
  double ss = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double g = G[(int64_t)bi * n + i];
    seed[i] = g;
    ss = fma(g, g, ss);
  }
  const double nrm = sqrt(block_sum(ss, red));
  for (int i = threadIdx.x; i < n; i += 256) seed[i] = seed[i] / nrm;
  __syncthreads();
  if ((int)blockIdx.x >= ny_blocks) {
    // x[j] = sum_i G_0[j, i] seed[i]      one wavefront per row
    const int j = ((int)blockIdx.x - ny_blocks) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j < n) {
      double s = 0.0;
      for (int i = lane; i < n; i += 64) s = fma(G0[(int64_t)j * n + i], seed[i], s);
      s = wave_sum(s);
      if (lane == 0) x[j] = s;
    }
    return;
  }
  // y[c] = sum_j M[j, c] seed[j]
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s = 0.0;
  if (c < k)
    for (int j = ry; j < n; j += 8) s = fma(M[(int64_t)j * k + c], seed[j], s);
  rsum[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < k) {
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) tot += rsum[g][cx];
    y[c] = tot;
  }
}

// F3: normalise x and y, apply the sign rule, write wA / wB / sigma / info   (1024 threads; x, y may
// live in global memory or in LDS)
// info for the caller: [converged within the budget, squarings computed]; [0, -1] when the one-launch chain gave up waiting for a
// workgroup that never became resident (gave: its flag word, nullptr for the launch-per-squaring form): the loadings are NaN
// and the caller must repeat the extraction through cmtfpls_rank1_launches_f64 / with the chain switched off
__device__ __forceinline__ void rank1_write_info(double* __restrict__ info, const Rank1Ctl* __restrict__ ctl, const double* gave) {
  if (!info) return;
  if (gave && !chain_empty(chain_ld(gave))) { info[0] = 0.0; info[1] = -1.0; return; }
  info[0] = (ctl->done || ctl->last_step >= 0) ? 1.0 : 0.0;
  info[1] = (double)ctl->steps_used;
}

__device__ __forceinline__ void rank1_final_body(const double* x, const double* y,
                                                 int n, int k, int x_is_A, const Rank1Ctl* __restrict__ ctl,
                                                 double* __restrict__ wA, double* __restrict__ wB,
                                                 double* __restrict__ sigma, double* __restrict__ info, const double* gave) {
  __shared__ double red[2][16];
  __shared__ double bestv[16];
  __shared__ int besti[16];
  double sx = 0.0, sy = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) sx = fma(x[i], x[i], sx);
  for (int i = threadIdx.x; i < k; i += 1024) sy = fma(y[i], y[i], sy);
  const double nx = sqrt(block_sum(sx, red[0]));
  const double ny = sqrt(block_sum(sy, red[1]));
  // sign rule on the LAST mode's vector wB: its largest-|.| entry is positive (first index on ties)
  const double* vb = x_is_A ? y : x;
  const int nb = x_is_A ? k : n;
  double bv = -1.0;
  int bi = 0;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    const double d = fabs(vb[i]);
    if (d > bv) { bv = d; bi = i; }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ov = __shfl_xor(bv, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bestv[threadIdx.x >> 6] = bv; besti[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = bestv[0];
  bi = besti[0];
  for (int w = 1; w < 16; ++w)
    if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
  // a chain that gave up may still have reached a usable final G (the give-up fell into its last step): the caller is told to
  // repeat the extraction either way, and every rank of a sharded fit must see that in the data -- the loadings are NaN then
  const bool gv = gave && !chain_empty(chain_ld(gave));
  const double sgn = gv ? __longlong_as_double(0x7FF8000000000000ll) : ((vb[bi] < 0.0) ? -1.0 : 1.0);
  double* ox = x_is_A ? wA : wB;
  double* oy = x_is_A ? wB : wA;
  for (int i = threadIdx.x; i < n; i += 1024) ox[i] = sgn * (x[i] / nx);
  for (int i = threadIdx.x; i < k; i += 1024) oy[i] = sgn * (y[i] / ny);
  if (threadIdx.x == 0) {
    // y = M^T seed has norm ~ sigma, x = M y has norm ~ sigma^2: sigma_1 = |x| / |y|
    if (sigma) sigma[0] = nx / ny;
    rank1_write_info(info, ctl, gave);
  }
}

__global__ __launch_bounds__(1024) void rank1_final_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                          int n, int k, int x_is_A, const Rank1Ctl* __restrict__ ctl,
                                                          double* __restrict__ wA, double* __restrict__ wB,
                                                          double* __restrict__ sigma, double* __restrict__ info, const double* gave) {
  rank1_final_body(x, y, n, k, x_is_A, ctl, wA, wB, sigma, info, gave);
}

// The last kernel of the extraction AND the score of a FEW LONG rows with the loading it has just formed (round 3): inside the
// cross-covariance loop the rank-1 extraction of Z is always followed by Y^T t = S (wA (x) wB) on the M rows of S (M <= 64 rows of
// >= 8192 elements: cmtfpls_score_s_f64 takes them one 1024-thread workgroup per row, score_fewrows_kernel) -- one launch of pure
// latency less per iteration.  Every workgroup normalises and sign-fixes the two vectors itself (a few hundred elements, into
// LDS; workgroup 0 also stores them and the flags), then takes its row exactly as score_fewrows_kernel does: same per-lane
// order, same block sum, the same bits.
__global__ __launch_bounds__(1024) void rank1_final_score_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                                int n, int k, int x_is_A, const Rank1Ctl* __restrict__ ctl,
                                                                double* __restrict__ wA, double* __restrict__ wB,
                                                                double* __restrict__ info, const double* __restrict__ S, int A, int B,
                                                                double* __restrict__ t, const double* gave) {
  extern __shared__ double lds[];
  __shared__ double red[2][16];
  __shared__ double bestv[16];
  __shared__ int besti[16];
  double* sA = lds;
  double* sB = lds + ((A + 1) & ~1);
  double sx = 0.0, sy = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) sx = fma(x[i], x[i], sx);
  for (int i = threadIdx.x; i < k; i += 1024) sy = fma(y[i], y[i], sy);
  const double nx = sqrt(block_sum(sx, red[0]));
  const double ny = sqrt(block_sum(sy, red[1]));
  const double* vb = x_is_A ? y : x;                      // sign rule as rank1_final_body: wB's largest-|.| entry is positive
  const int nb = x_is_A ? k : n;
  double bv = -1.0;
  int bi = 0;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    const double d = fabs(vb[i]);
    if (d > bv) { bv = d; bi = i; }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ov = __shfl_xor(bv, m, 64);
    const int oi = __shfl_xor(bi, m, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bestv[threadIdx.x >> 6] = bv; besti[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = bestv[0];
  bi = besti[0];
  for (int w = 1; w < 16; ++w)
    if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
  const bool gv = gave && !chain_empty(chain_ld(gave));    // (as rank1_final_body: NaN loadings when the chain gave up)
  const double sgn = gv ? __longlong_as_double(0x7FF8000000000000ll) : ((vb[bi] < 0.0) ? -1.0 : 1.0);
  double* lx = x_is_A ? sA : sB;
  double* ly = x_is_A ? sB : sA;
  double* ox = x_is_A ? wA : wB;
  double* oy = x_is_A ? wB : wA;
  const bool store = blockIdx.x == 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double v = sgn * (x[i] / nx);
    lx[i] = v;
    if (store) ox[i] = v;
  }
  for (int i = threadIdx.x; i < k; i += 1024) {
    const double v = sgn * (y[i] / ny);
    ly[i] = v;
    if (store) oy[i] = v;
  }
  if (store && threadIdx.x == 0) rank1_write_info(info, ctl, gave);
  __syncthreads();
  // the row, as score_fewrows_kernel<double, false>
  using VT = Pack<double, 2>;
  const int64_t P = (int64_t)A * B;
  const int64_t step = (int64_t)1024 * 2;
  const double* __restrict__ xr = S + (int64_t)blockIdx.x * P;
  KronWalk w((int64_t)threadIdx.x * 2, step, B);
  double acc = 0.0;
  int64_t c = (int64_t)threadIdx.x * 2;
  constexpr int UN = 4;
  for (; c + (UN - 1) * step < P; c += UN * step) {
    VT xs[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) xs[u] = ld_stream(reinterpret_cast<const VT*>(xr + c + u * step));
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      double d = 0.0;
      d = fma(xs[u].e[0], sB[w.k], d);
      d = fma(xs[u].e[1], sB[w.k + 1], d);
      acc = fma(sA[w.j], d, acc);
      w.next();
    }
  }
  for (; c < P; c += step) {
    const VT xv = ld_stream(reinterpret_cast<const VT*>(xr + c));
    double d = 0.0;
    d = fma(xv.e[0], sB[w.k], d);
    d = fma(xv.e[1], sB[w.k + 1], d);
    acc = fma(sA[w.j], d, acc);
    w.next();
  }
  acc = block_sum(acc, red[0]);
  if (threadIdx.x == 0) t[blockIdx.x] = acc;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// The one-launch chain needs every one of its workgroups resident at once.  On a GPU this process has to itself (one process
// per GPU: the deployment) that holds; where it does not (several processes on one card) an extraction reports info = [0, -1] and the
// caller switches the chain off for the rest of the process: cmtfpls_rank1_chain_enable(0).
static std::atomic<int> g_chain_on{1};

// workgroups of the chain kernel the device can hold at once: two per CU (198 VGPRs, 50 KB of LDS)
static int chain_slots() {
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slots = 2 * cus;
  }
  return slots;
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {

size_t cmtfpls_rank1_workspace_bytes(int A, int B) {
  if (A <= 0 || B <= 0) return 0;
  const size_t n = (size_t)(A < B ? A : B), k = (size_t)(A < B ? B : A);
  const size_t nt = (n + kTile - 1) / kTile;
  const size_t chain = nt * nt <= 256 ? align_up(((size_t)(kChainMaxSteps + 1) * (n * n + nt + nt * nt) + 1) * sizeof(double), 256) : 0;
  return align_up(rank1_ctl_bytes((int)nt), 256) + 3 * align_up(n * n * sizeof(double), 256) +
         align_up((size_t)A * B * sizeof(double), 256) + align_up(n * sizeof(double), 256) + align_up(k * sizeof(double), 256) + chain;
}

static int rank1_run(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                     int n_squarings, void* ws, size_t ws_bytes, void* stream, const double* S, int M, double* tq, bool allow_chain = true);

int cmtfpls_rank1_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                      int n_squarings, void* ws, size_t ws_bytes, void* stream) {
  return rank1_run(Z, A, B, wA, wB, sigma, info, n_squarings, ws, ws_bytes, stream, nullptr, 0, nullptr);
}

#ifdef CMTFPLS_CHAIN_PROFILE
int cmtfpls_debug_chain_prof(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_prof), sizeof(unsigned long long) * 1024); }
#endif
void cmtfpls_rank1_chain_enable(int on) { g_chain_on.store((on == 1 || on == 2) ? on : 0, std::memory_order_relaxed); }
int cmtfpls_rank1_chain_enabled(void) { return g_chain_on.load(std::memory_order_relaxed); }

int cmtfpls_rank1_launches_f64(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                               int n_squarings, void* ws, size_t ws_bytes, void* stream) {
  return rank1_run(Z, A, B, wA, wB, sigma, info, n_squarings, ws, ws_bytes, stream, nullptr, 0, nullptr, false);
}

int cmtfpls_rank1_score_f64(const double* Z, int A, int B, double* wA, double* wB, double* info, int n_squarings,
                            const double* S, int M, double* tq, void* ws, size_t ws_bytes, void* stream) {
  if (!S || !tq || M <= 0) { set_error("rank1_score: bad argument"); return CMTFPLS_EINVAL; }
  const int64_t P = (int64_t)A * B;
  if (M > 64 || P < 8192 || (B % 2) != 0 || ((size_t)(((A + 1) & ~1) + B) * sizeof(double)) > 60 * 1024 ||
      (reinterpret_cast<uintptr_t>(S) & 15) != 0) {       // outside the few-long-rows form: the two entries, one after the other
    int rc = cmtfpls_rank1_f64(Z, A, B, wA, wB, nullptr, info, n_squarings, ws, ws_bytes, stream);
    if (rc == CMTFPLS_OK) rc = cmtfpls_score_s_f64(S, M, A, B, wA, wB, tq, stream);
    return rc;
  }
  return rank1_run(Z, A, B, wA, wB, nullptr, info, n_squarings, ws, ws_bytes, stream, S, M, tq);
}

static int rank1_run(const double* Z, int A, int B, double* wA, double* wB, double* sigma, double* info,
                     int n_squarings, void* ws, size_t ws_bytes, void* stream, const double* S, int M, double* tq, bool allow_chain) {
  if (!Z || !wA || !wB || A <= 0 || B <= 0) { set_error("rank1: bad argument"); return CMTFPLS_EINVAL; }
  const int n = A < B ? A : B, k = A < B ? B : A;
  if (n > kTile * kMaxTiles) { set_error("rank1: min(A, B) > 4096 unsupported"); return CMTFPLS_EUNSUPPORTED; }
  if ((size_t)n * sizeof(double) > 64 * 1024) { set_error("rank1: seed exceeds the LDS"); return CMTFPLS_EUNSUPPORTED; }
  if (n_squarings < 1) n_squarings = 1;
  if (n_squarings > kMaxSteps) n_squarings = kMaxSteps;
  if (!ws || ws_bytes < cmtfpls_rank1_workspace_bytes(A, B)) { set_error("rank1: workspace too small"); return CMTFPLS_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  char* p = static_cast<char*>(ws);
  Rank1Ctl* ctl = reinterpret_cast<Rank1Ctl*>(p);
  p += align_up(rank1_ctl_bytes((n + kTile - 1) / kTile), 256);
  double* buf0 = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * n * sizeof(double), 256);
  double* buf1 = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * n * sizeof(double), 256);
  double* g0keep = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * n * sizeof(double), 256);
  double* Zt = reinterpret_cast<double*>(p);
  p += align_up((size_t)A * B * sizeof(double), 256);
  double* xv = reinterpret_cast<double*>(p);
  p += align_up((size_t)n * sizeof(double), 256);
  double* yv = reinterpret_cast<double*>(p);
  p += align_up((size_t)k * sizeof(double), 256);
  double* chain_region = reinterpret_cast<double*>(p);

  const double* M0 = Z;   // n x k with n on the smaller side
  if (A > B) {
    const int64_t tot = (int64_t)A * B;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Z, A, B, Zt);
    M0 = Zt;
  }
  const int nt = (n + kTile - 1) / kTile;
  const dim3 grid(nt, nt), block(kTile, kTile);
  const int ny_blocks = (k + 31) / 32;
  const double* gave = nullptr;
  if (allow_chain && g_chain_on.load(std::memory_order_relaxed) && nt * nt <= 256 && nt * nt <= chain_slots() && n_squarings <= kChainMaxSteps) {
    // every step in ONE launch (syrk_chain_kernel): G_s | traces | sums of squares of steps 0 .. n_squarings, preset to the
    // pattern that means "not yet written"
    const size_t nn = (size_t)n * n, steps = (size_t)n_squarings + 1;
    double* gbufs = chain_region;
    double* trs = gbufs + steps * nn;
    double* fros = trs + steps * nt;
    double* gave_up = fros + steps * (size_t)nt * nt;       // stays at the preset unless a wavefront gives up
    gave = gave_up;
    const int64_t words = (int64_t)(steps * (nn + nt + (size_t)nt * nt) + 1);
    hipLaunchKernelGGL(chain_preset_kernel, dim3((unsigned)((words + 2047) / 2048)), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(chain_region), words);
    // (mode 2, tests only: the last row of workgroups is not launched, so its partners wait in vain and the give-up path runs)
    const dim3 cgrid(nt, (g_chain_on.load(std::memory_order_relaxed) == 2 && nt > 1) ? nt - 1 : nt);
    hipLaunchKernelGGL(syrk_chain_kernel, cgrid, block, 0, st, M0, n, k, gbufs, trs, fros, gave_up, ctl, n_squarings);
    hipLaunchKernelGGL(rank1_seed_xy_kernel, dim3(ny_blocks + (n + 3) / 4), dim3(256), (size_t)n * sizeof(double), st,
                       M0, n, k, gbufs, (int64_t)nn, gbufs, ctl, n_squarings, ny_blocks, yv, xv);
  } else {
    // step 0: G_0 = M0 M0^T -> buf0 ; step s: G_s = scale^2 G_{s-1} G_{s-1}^T -> buf[s & 1]
    hipLaunchKernelGGL(syrk_step_kernel, grid, block, 0, st, M0, n, k, k, buf0, ctl, 0, 0, g0keep);
    for (int s = 1; s <= n_squarings; ++s) {
      const double* in = (s & 1) ? buf0 : buf1;
      double* out = (s & 1) ? buf1 : buf0;
      hipLaunchKernelGGL(syrk_step_kernel, grid, block, 0, st, in, n, n, n, out, ctl, s, s & 1, (double*)nullptr);
    }
    // (a single-workgroup fusion of the finish kernels was measured in round 1: no faster than parallel launches)
    hipLaunchKernelGGL(rank1_seed_xy_kernel, dim3(ny_blocks + (n + 3) / 4), dim3(256), (size_t)n * sizeof(double), st,
                       M0, n, k, buf0, (int64_t)(buf1 - buf0), g0keep, ctl, n_squarings & 1, ny_blocks, yv, xv);
  }
  if (S) {
    const size_t lds = (size_t)(((A + 1) & ~1) + B) * sizeof(double);
    hipLaunchKernelGGL(rank1_final_score_kernel, dim3((unsigned)M), dim3(1024), lds, st, xv, yv, n, k, (A <= B) ? 1 : 0, ctl, wA, wB, info,
                       S, A, B, tq, gave);
    return check_launch("rank1_score");
  }
  hipLaunchKernelGGL(rank1_final_kernel, dim3(1), dim3(1024), 0, st, xv, yv, n, k, (A <= B) ? 1 : 0, ctl, wA, wB, sigma, info, gave);
  return check_launch("rank1");
}

}  // extern "C"
