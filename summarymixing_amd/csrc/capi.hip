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

namespace smx {
const uint64_t* g_step_counter = nullptr;
__global__ void step_counter_add_kernel(uint64_t* c, uint64_t inc) { c[0] += inc; }
}  // namespace smx

extern "C" int smx_set_step_counter(const uint64_t* dev_counter) {
  smx::g_step_counter = dev_counter;
  return SMX_OK;
}
extern "C" int smx_step_counter_add(uint64_t* dev_counter, uint64_t inc, void* stream) {
  SMX_REQUIRE(dev_counter, "smx_step_counter_add: null pointer");
  hipLaunchKernelGGL(smx::step_counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), dev_counter, inc);
  return smx::check_launch("smx_step_counter_add");
}

extern "C" int smx_version(void) { return SMX_VERSION; }
extern "C" const char* smx_last_error(void) { return smx::last_error_buf(); }
