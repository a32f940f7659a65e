// Device side of the reference's synthetic-data recipe (cmtf_pls/synthetic.py:59-74 and :5-34): the dense CP
// tensor of the drawn factors is cmtfpls_recon_* (recon.hip); this file adds the N(0, error) noise
// (synthetic.py:71,74) and, for BASELINE configs[3], the i.i.d. NaN mask -- in place, from a COUNTER-BASED
// generator (Philox4x32-10 keyed by the seed, counter = global element index / 4), so that element (i, c) of the
// tensor gets the same noise whichever rank forms the rows around it: rank g's shard of a sharded run IS rows
// [g I/G, (g+1) I/G) of the single-GPU tensor, noise included.  (NumPy's PCG64 stream of the host recipe cannot
// be reproduced on the device; the factors -- the part of the recipe that defines the problem -- are drawn on the
// host with the reference's generator and order.)
#include "common.hpp"

namespace cmtfpls {

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t counter, uint32_t stream, uint64_t key) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = stream, c3 = 0u;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ double unit_open(uint32_t x) { return ((double)x + 0.5) * 2.3283064365386963e-10; }   // (0, 1)

// four standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ void normals4(uint64_t counter, uint64_t key, double out[4]) {
  const Philox4 p = philox4x32_10(counter, 0u, key);
  const double r0 = sqrt(-2.0 * log(unit_open(p.v[0]))), r1 = sqrt(-2.0 * log(unit_open(p.v[2])));
  double s0, c0, s1, c1;
  sincospi(2.0 * unit_open(p.v[1]), &s0, &c0);
  sincospi(2.0 * unit_open(p.v[3]), &s1, &c1);
  out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

// X[e] += sigma * N(0,1)[offset + e];  then X[e] = NaN where U(0,1)[offset + e] < nan_fraction
template <typename T>
__global__ __launch_bounds__(256) void add_noise_kernel(T* __restrict__ X, int64_t n, int64_t nquads, double sigma, uint64_t seed,
                                                       uint64_t offset, double nan_fraction) {
  // thread = one GLOBAL quad of the stream at a time (4 consecutive global elements share one Philox block), grid
  // stride over the quads; a buffer that starts or ends inside a quad uses only its own elements of it
  const uint64_t q0 = offset >> 2;
  for (int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x; qi < nquads; qi += (int64_t)gridDim.x * 256) {
    const uint64_t gq = q0 + (uint64_t)qi;
    const int64_t e0 = (int64_t)(gq * 4 - offset);              // local index of the quad's first element (may be < 0)
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    if (sigma != 0.0) normals4(gq, seed, z);
    Philox4 m{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
    if (nan_fraction > 0.0) m = philox4x32_10(gq, 1u, seed);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t e = e0 + k;
      if (e >= 0 && e < n) {
        T v = (T)((double)X[e] + sigma * z[k]);
        if (nan_fraction > 0.0 && unit_open(m.v[k]) < nan_fraction) v = (T)NAN;
        X[e] = v;
      }
    }
  }
}

template <typename T>
static int run_add_noise(T* X, int64_t n, double sigma, uint64_t seed, uint64_t offset, double nan_fraction, hipStream_t st) {
  if (!X || n <= 0 || !(sigma >= 0.0) || !(nan_fraction >= 0.0) || nan_fraction > 1.0) { set_error("add_noise: bad argument"); return CMTFPLS_EINVAL; }
  const int64_t quads = (int64_t)(((offset + (uint64_t)n + 3) >> 2) - (offset >> 2));
  int64_t blocks = (quads + 255) / 256;
  if (blocks > 65536) blocks = 65536;                   // grid stride beyond (a launch is limited to < 2^32 threads)
  hipLaunchKernelGGL((add_noise_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, X, n, quads, sigma, seed, offset, nan_fraction);
  return check_launch("add_noise");
}

}  // namespace cmtfpls

using namespace cmtfpls;

extern "C" {
int cmtfpls_add_noise_f32(float* X, int64_t n, double sigma, uint64_t seed, uint64_t offset, double nan_fraction, void* stream) {
  return run_add_noise<float>(X, n, sigma, seed, offset, nan_fraction, (hipStream_t)stream);
}
int cmtfpls_add_noise_f64(double* X, int64_t n, double sigma, uint64_t seed, uint64_t offset, double nan_fraction, void* stream) {
  return run_add_noise<double>(X, n, sigma, seed, offset, nan_fraction, (hipStream_t)stream);
}
}
