// capi.hip — error plumbing and version of the C-ABI (include/smx.h).
#include <stdarg.h>

#include "smx_common.h"
#include "gemm_common.h"

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

#include <stdlib.h>

namespace smx {
static int env_i(const char* n, int dflt) { const char* e = getenv(n); return (e && e[0]) ? atoi(e) : dflt; }
const smx_config& cfg() {
  // function-local static: initialised exactly once (thread-safe), on the first call of any entry point that needs a knob
  static const smx_config c = [] {
    smx_config k;
    memset(&k, 0, sizeof(k));
    k.ln_tile_rows = 128;
    k.t256 = env_i("SMX_T256", 1);
    k.panel_rows = env_i("SMX_PANEL_ROWS", 0);
    k.pool_fuse_max_rows = env_i("SMX_POOL_FUSE_MAX_ROWS", 16384);
    k.ln_tile64 = env_i("SMX_LN_TILE64", 1);
#ifdef SMX_DIAG
    k.gemm_ablate = env_i("SMX_GEMM_ABLATE", 0);
    k.wgroup_ablate = env_i("SMX_WGROUP_ABLATE", 0);
    k.dwroll_ablate = env_i("SMX_DWROLL_ABLATE", 0);
    k.diag_build = 1;
#endif
    return k;
  }();
  return c;
}
__global__ void step_counter_add_kernel(uint64_t* c, uint64_t inc) { c[0] += inc; }
}  // namespace smx

extern "C" int smx_step_counter_add(uint64_t* dev_counter, uint64_t inc, void* stream) {
  SMX_REQUIRE(dev_counter, "smx_step_counter_add: null pointer");
  hipLaunchKernelGGL(smx::step_counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), dev_counter, inc);
  return smx::check_launch("smx_step_counter_add");
}

extern "C" int smx_stream_capture_id(void* stream, uint64_t* id) {
  SMX_REQUIRE(id, "smx_stream_capture_id: null pointer");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  const hipError_t err = hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &st, &cid);
  if (err != hipSuccess) return smx::fail(SMX_ELAUNCH, "smx_stream_capture_id: %s", hipGetErrorString(err));
  *id = st == hipStreamCaptureStatusActive ? (uint64_t)cid : 0;
  return SMX_OK;
}

extern "C" int smx_get_config(smx_config* out) {
  SMX_REQUIRE(out, "smx_get_config: null pointer");
  *out = smx::cfg();
  return SMX_OK;
}
extern "C" int smx_gemm_ln_tile_rows(void) { return smx::cfg().ln_tile_rows; }
extern "C" int smx_gemm_ln_tile_rows_for(int N, int M) { return smx::ln_tile_rows_for(N, M); }
extern "C" int smx_version(void) { return SMX_VERSION; }
extern "C" const char* smx_last_error(void) { return smx::last_error_buf(); }
