// capi.hip — error plumbing and version of the C-ABI (include/smx.h).
#include <stdarg.h>

#include "smx_common.h"

namespace smx {

static thread_local char g_err[512] = "";

char* last_error_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SMX_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return SMX_OK;
}

}  // namespace smx

extern "C" int smx_version(void) { return SMX_VERSION; }
extern "C" const char* smx_last_error(void) { return smx::last_error_buf(); }
