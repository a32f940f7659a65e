// Thin RCCL wrappers: the all-reduce(sum) the sharded NIPALS loop issues -- Z (P doubles) and
// Y^T t (M doubles) per iteration, T^T [T | u] per component -- for callers that drive the C ABI without torch.
// (cmtf_pls_amd itself issues the same collectives through torch.distributed, whose backend "nccl" is RCCL.)
//
// libcmtfpls does NOT link RCCL: the process that owns a communicator has already loaded the RCCL it came from, and
// two RCCL builds in one process (e.g. torch's bundled one and /opt/rocm's) must not be mixed.  ncclAllReduce is
// resolved at first use from the symbols already visible in the process (RTLD_DEFAULT), else from librccl.so.1 /
// librccl.so on the loader path; CMTFPLS_EUNSUPPORTED when neither is there.
#include <dlfcn.h>

#include "common.hpp"

namespace cmtfpls {

// ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;      // rccl.h: ncclSum, ncclFloat32, ncclFloat64

static nccl_allreduce_fn resolve_allreduce() {
  static nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!sym) {
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) sym = dlsym(h, "ncclAllReduce");
    }
    return reinterpret_cast<nccl_allreduce_fn>(sym);
  }();
  return fn;
}

static int allreduce_sum(void* comm, void* buf, size_t count, int dtype, void* stream) {
  if (!comm || !buf) { set_error("allreduce_sum: bad argument"); return CMTFPLS_EINVAL; }
  if (count == 0) return CMTFPLS_OK;
  nccl_allreduce_fn fn = resolve_allreduce();
  if (!fn) { set_error("allreduce_sum: no RCCL in this process (ncclAllReduce not found, librccl.so not loadable)"); return CMTFPLS_EUNSUPPORTED; }
  const int rc = fn(buf, buf, count, dtype, kNcclSum, comm, (hipStream_t)stream);
  if (rc != 0) {
    char msg[96];
    snprintf(msg, sizeof(msg), "allreduce_sum: ncclAllReduce returned %d", rc);
    set_error(msg);
    return CMTFPLS_EHIP;
  }
  return CMTFPLS_OK;
}

}  // namespace cmtfpls

extern "C" {
int cmtfpls_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream) {
  return cmtfpls::allreduce_sum(comm, buf, count, cmtfpls::kNcclFloat64, stream);
}
int cmtfpls_allreduce_sum_f32(void* comm, float* buf, size_t count, void* stream) {
  return cmtfpls::allreduce_sum(comm, buf, count, cmtfpls::kNcclFloat32, stream);
}
}
