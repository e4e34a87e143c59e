// gemm_ln512.hip — the LayerNorm-fused instantiations of gemm_kernel (gemm_kernel.h) on the row-complete 128 x 512 tile (one workgroup per CU, the software-pipelined main loop of the 256 x 256 tile; d_model = 512)
// and the same tile without a LayerNorm (SMX_T256=3: experiments / tests).
// A translation unit of its own so that it compiles next to gemm.hip (each of these kernels keeps 256-512 registers per lane
// and takes hipcc 10-20 s).
#include "gemm_kernel.h"

namespace smx {

template <bool B_KC>
static int launch_ln(GemmParams& p, hipStream_t s) {
  typedef bf16_t T;
  p.tiles_n = (p.N + 127) / 128;
  p.tiles_m = 1;
  const dim3 grid(p.tiles_n), block(256);
  const bool lnb = (p.e.flags & SMX_EPI_LN_BWD) != 0, lnf = (p.e.flags & SMX_EPI_LN_FWD) != 0, xf32 = (p.e.io_flags & SMX_IO_LNX_F32) != 0;
  const bool ext = lnb && (p.e.lnf_act != SMX_ACT_NONE || p.e.z);
  if (plan_only(p, 0, true, B_KC, 128, 512, true, ext ? (xf32 ? 7 : 3) : (lnb ? (xf32 ? 5 : 1) : (lnf ? 2 : 0)), 0)) return SMX_OK;
  if (ext && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true, 7>), grid, block, 0, s, p);
  else if (ext) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true, 3>), grid, block, 0, s, p);
  else if (lnb && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true, 5>), grid, block, 0, s, p);
  else if (lnb) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true, 1>), grid, block, 0, s, p);
  else if (lnf) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true, 2>), grid, block, 0, s, p);
  else {  // no LayerNorm: the plain epilogues on the same tile (SMX_T256=3)
    hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 512, true>), grid, block, 0, s, p);
    if (p.e.colsum) launch_colsum_partials(reinterpret_cast<const float*>(p.e.workspace), p.tiles_n, p.M, p.e.colsum, s);
  }
  return check_launch("smx_gemm");
}

// p: a bf16 NT (b_kc) / NN GEMM with M == 512, whole aligned vectors, splits == 1, batch == 1 (checked by launch_layout, gemm.hip)
int launch_ln_fused_512(GemmParams& p, bool b_kc, hipStream_t s) { return b_kc ? launch_ln<true>(p, s) : launch_ln<false>(p, s); }

}  // namespace smx
