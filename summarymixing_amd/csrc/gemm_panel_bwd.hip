// gemm_panel_bwd.hip — the act-grad instantiations of the panel-resident GEMM (gemm_panel.h): dZ = D(dY W * act'(Z)).
// A translation unit of its own so that it compiles next to gemm_panel.hip.
#include "gemm_panel.h"

namespace smx {

int launch_panel_actgrad(const PanelParams& p, int K, int act, hipStream_t s) { return launch_panel_mode<1>(p, K, act, s); }

}  // namespace smx
