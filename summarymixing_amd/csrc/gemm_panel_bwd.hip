// gemm_panel_bwd.hip — the act-grad instantiations of the panel-resident GEMM (gemm_panel.h): dZ = D(dY W * act'(Z)).
// A translation unit of its own so that it compiles next to gemm_panel.hip.
#include "gemm_panel.h"

namespace smx {

int launch_panel_actgrad(const PanelParams& p, int K, int act, hipStream_t s) { return launch_panel_mode<1>(p, K, act, s); }

// split-K slabs (MODE 2): one instantiation per K and panel height
int launch_panel_slabs(const PanelParams& p, int K, hipStream_t s) {
  const int rows = p.rows;
  const dim3 grid(((p.N + rows - 1) / rows) * p.csplit, p.nslice), block(512);
#define SMX_SLAB_CASE(KK, RR) if (K == KK && rows == RR) { hipLaunchKernelGGL((gemm_panel_kernel<KK, 2, SMX_ACT_NONE, RR>), grid, block, 0, s, p); return check_launch("smx_gemm_panel_slabs"); }
  SMX_SLAB_CASE(256, 128) SMX_SLAB_CASE(256, 64) SMX_SLAB_CASE(256, 32) SMX_SLAB_CASE(512, 64) SMX_SLAB_CASE(512, 32)   // (K = 512 x 128 rows + the 64 KB of float32 scratch would pass the CU's 160 KB)
#undef SMX_SLAB_CASE
  return fail(SMX_EUNSUPPORTED, "smx_gemm_panel_slabs: K = %d has no instantiation", K);
}

}  // namespace smx
