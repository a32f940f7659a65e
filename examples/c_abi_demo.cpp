// Torch-free use of libcmtfpls through its C ABI (include/cmtfpls.h): one NIPALS iteration and a
// deflation on a small order-3 tensor, with plain hipMalloc buffers, checked against host loops.
// Build: cmtf_pls_amd/csrc/build.sh (-> examples/c_abi_demo).   Run: examples/c_abi_demo   (exit 0 = ok)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../include/cmtfpls.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ABI(x) do { int rc_ = (x); if (rc_ != CMTFPLS_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, cmtfpls_last_error()); return 3; } } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
template <typename T>
static std::vector<T> to_host(const T* d, size_t n) {
  std::vector<T> h(n);
  (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}
static double maxdiff(const std::vector<double>& a, const std::vector<double>& b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i) m = fmax(m, fabs(a[i] - b[i]));
  return m;
}

// The collectives of the sharded loop without torch: a ONE-rank RCCL communicator (what one GPU can run), the Z of the
// iteration above all-reduced in place through cmtfpls_allreduce_sum_f64.  RCCL is opened at run time (so this demo
// links no RCCL either); with ranks on several GPUs the call is the same and sums the ranks' partial Z.
// Returns 0 ok, 1 failed, -1 RCCL not usable here (reported, not an error of the C ABI).
static int rccl_one_rank_allreduce(double* dZ, int64_t P, hipStream_t st) {
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { printf("rccl: library not found (%s)\n", dlerror()); return -1; }
  struct UniqueId { char internal[128]; };                                     // rccl.h: NCCL_UNIQUE_ID_BYTES
  typedef int (*get_id_fn)(UniqueId*);
  typedef int (*init_fn)(void**, int, UniqueId, int);
  typedef int (*destroy_fn)(void*);
  get_id_fn get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
  init_fn init = (init_fn)dlsym(h, "ncclCommInitRank");
  destroy_fn destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
  if (!get_id || !init || !destroy) { printf("rccl: symbols missing\n"); return -1; }
  UniqueId id;
  void* comm = nullptr;
  int rc = get_id(&id);
  if (rc == 0) rc = init(&comm, 1, id, 0);
  if (rc != 0 || !comm) { printf("rccl: one-rank communicator not available here (ncclResult %d)\n", rc); return -1; }
  std::vector<double> before = to_host(dZ, (size_t)P);
  const int arc = cmtfpls_allreduce_sum_f64(comm, dZ, (size_t)P, st);
  if (arc != CMTFPLS_OK) { fprintf(stderr, "cmtfpls_allreduce_sum_f64 -> %d (%s)\n", arc, cmtfpls_last_error()); destroy(comm); return 1; }
  if (hipStreamSynchronize(st) != hipSuccess) { destroy(comm); return 1; }
  std::vector<double> after = to_host(dZ, (size_t)P);
  destroy(comm);
  for (int64_t p = 0; p < P; ++p)
    if (after[p] != before[p]) { printf("rccl: one-rank all-reduce changed element %lld\n", (long long)p); return 1; }
  printf("rccl: one-rank all-reduce(sum) of Z (%lld doubles) through cmtfpls_allreduce_sum_f64 OK\n", (long long)P);
  return 0;
}

int main() {
  const int64_t I = 300;
  const int J = 12, K = 8, M = 3;
  const int64_t P = (int64_t)J * K;
  std::vector<float> X(I * P);
  std::vector<double> Y(I * M), u(I);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0 - 0.5; };
  // rank-one signal + noise so that the leading singular pair of Z is well separated
  std::vector<double> a(I), b(J), c(K);
  for (auto& v : a) v = rnd();
  for (auto& v : b) v = rnd();
  for (auto& v : c) v = rnd();
  for (int64_t i = 0; i < I; ++i) {
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) X[i * P + j * K + k] = (float)(8.0 * a[i] * b[j] * c[k] + 0.1 * rnd());
    for (int m = 0; m < M; ++m) Y[i * M + m] = a[i] * (m + 1) + 0.05 * rnd();
    u[i] = Y[i * M];
  }
  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));
  float* dX = to_device(X);
  double *dY = to_device(Y), *du = to_device(u);
  double *dZ, *dwA, *dwB, *dt, *dq, *dpart, *dssq;
  HIPCHECK(hipMalloc(&dZ, P * 8));
  HIPCHECK(hipMalloc(&dwA, J * 8));
  HIPCHECK(hipMalloc(&dwB, K * 8));
  HIPCHECK(hipMalloc(&dt, I * 8));
  HIPCHECK(hipMalloc(&dq, M * 8));
  HIPCHECK(hipMalloc(&dpart, cmtfpls_sweep_partials() * 8));
  HIPCHECK(hipMalloc(&dssq, 8));
  size_t wsn = cmtfpls_mode0_contract_workspace_bytes(I, P);
  size_t wr = cmtfpls_rank1_workspace_bytes(J, K), wsm = cmtfpls_small_workspace_bytes();
  void *ws1, *ws2, *ws3;
  HIPCHECK(hipMalloc(&ws1, wsn));
  HIPCHECK(hipMalloc(&ws2, wr));
  HIPCHECK(hipMalloc(&ws3, wsm));

  // tpls.py:83   Z = einsum(X, u)
  ABI(cmtfpls_mode0_contract_f32(dX, I, P, du, dZ, 0, ws1, wsn, st));
  // tpls.py:86   leading singular pair of Z
  ABI(cmtfpls_rank1_f64(dZ, J, K, dwA, dwB, nullptr, nullptr, 30, ws2, wr, st));
  // tpls.py:97   t = multi_mode_dot(X, [wJ, wK])
  ABI(cmtfpls_score_f32(dX, I, J, K, dwA, dwB, nullptr, dt, st));
  // tpls.py:100-101   q = Y^T t / |.|
  ABI(cmtfpls_gram_tn_f64(dY, M, M, dt, 1, 1, I, dq, ws3, wsm, st));
  ABI(cmtfpls_normalize_f64(dq, M, nullptr, st));
  // tpls.py:109   X -= outer(t, wJ, wK)  (+ |X|^2 of the result)
  ABI(cmtfpls_deflate_f32(dX, I, J, K, dt, dwA, dwB, dpart, st));
  ABI(cmtfpls_sum_f64(dpart, cmtfpls_sweep_partials(), dssq, st));
  HIPCHECK(hipStreamSynchronize(st));

  // host check
  std::vector<double> Zh(P, 0.0);
  for (int64_t i = 0; i < I; ++i)
    for (int64_t p = 0; p < P; ++p) Zh[p] += (double)X[i * P + p] * u[i];
  std::vector<double> wA = to_host(dwA, J), wB = to_host(dwB, K), Zd = to_host(dZ, P), td = to_host(dt, I), qd = to_host(dq, M);
  double e_z = maxdiff(Zh, Zd);
  // singular pair: Z wB = sigma wA and Z^T wA = sigma wB, unit norms
  double sigma = 0, na = 0, nb = 0, res = 0;
  for (int j = 0; j < J; ++j) { double r = 0; for (int k = 0; k < K; ++k) r += Zh[j * K + k] * wB[k]; sigma += r * wA[j]; na += wA[j] * wA[j]; }
  for (int k = 0; k < K; ++k) nb += wB[k] * wB[k];
  for (int j = 0; j < J; ++j) { double r = 0; for (int k = 0; k < K; ++k) r += Zh[j * K + k] * wB[k]; res = fmax(res, fabs(r - sigma * wA[j])); }
  std::vector<double> th(I, 0.0), qh(M, 0.0);
  for (int64_t i = 0; i < I; ++i)
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) th[i] += (double)X[i * P + j * K + k] * wA[j] * wB[k];
  double e_t = maxdiff(th, td), qn = 0;
  for (int64_t i = 0; i < I; ++i)
    for (int m = 0; m < M; ++m) qh[m] += Y[i * M + m] * th[i];
  for (int m = 0; m < M; ++m) qn += qh[m] * qh[m];
  for (int m = 0; m < M; ++m) qh[m] /= sqrt(qn);
  double e_q = maxdiff(qh, qd);
  std::vector<float> Xd = to_host(dX, (size_t)(I * P));
  double e_x = 0, ssq_h = 0;
  for (int64_t i = 0; i < I; ++i)
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) {
        const float want = (float)((double)X[i * P + j * K + k] - th[i] * wA[j] * wB[k]);
        e_x = fmax(e_x, fabs((double)want - (double)Xd[i * P + j * K + k]));
        ssq_h += (double)Xd[i * P + j * K + k] * Xd[i * P + j * K + k];
      }
  double ssq_d = to_host(dssq, 1)[0];
  printf("abi %d | dZ %.2e | sigma %.4f |wA|^2 %.15f |wB|^2 %.15f resid %.2e | dt %.2e | dq %.2e | dX %.2e | ssq rel %.2e\n",
         cmtfpls_abi_version(), e_z, sigma, na, nb, res / fabs(sigma), e_t, e_q, e_x, fabs(ssq_d - ssq_h) / ssq_h);
  const bool ok = e_z < 1e-9 && fabs(na - 1) < 1e-12 && fabs(nb - 1) < 1e-12 && res / fabs(sigma) < 1e-9 && sigma > 0 &&
                  e_t < 1e-9 && e_q < 1e-10 && e_x < 1e-6 && fabs(ssq_d - ssq_h) / ssq_h < 1e-10;
  const int coll = rccl_one_rank_allreduce(dZ, P, st);
  printf((ok && coll != 1) ? "C ABI demo OK\n" : "C ABI demo FAILED\n");
  return (ok && coll != 1) ? 0 : 1;
}
