// Error plumbing shared by every entry point of libswinvox_hip.so.
#include "common.h"
#include <stdarg.h>

namespace sv {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SV_ERR_LAUNCH;
  }
  return SV_OK;
}
}  // namespace sv

extern "C" const char* sv_last_error(void) { return sv::g_err; }
extern "C" int sv_version(void) { return 1; }
