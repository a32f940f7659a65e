// Error reporting and ABI version of libcmtfpls.
#include <string.h>

#include "common.hpp"

namespace cmtfpls {

static thread_local char g_err[256] = "";

void set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return CMTFPLS_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return CMTFPLS_EHIP;
}

}  // namespace cmtfpls

extern "C" {
int cmtfpls_abi_version(void) { return 1; }
const char* cmtfpls_last_error(void) { return cmtfpls::g_err; }
int cmtfpls_clear_error(void) {
  // a failed stream capture (a collective whose backend cannot be captured, ...) leaves the runtime's per-thread "last error" set;
  // every entry of this library reports hipGetLastError() after its launches and would blame the next, innocent, launch
  const hipError_t e = hipGetLastError();
  cmtfpls::g_err[0] = 0;
  return e == hipSuccess ? 0 : 1;
}

int cmtfpls_status_to_host(const void* src, void* dst_host, size_t bytes, void* event, void* stream) {
  if (!src || !dst_host || bytes == 0) { cmtfpls::set_error("status_to_host: bad argument"); return CMTFPLS_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && event) e = hipEventRecord((hipEvent_t)event, st);
  if (e != hipSuccess) { cmtfpls::set_error(hipGetErrorString(e)); return CMTFPLS_EHIP; }
  return CMTFPLS_OK;
}
}
