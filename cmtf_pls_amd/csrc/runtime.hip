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
}
