// gemm_ln256r64.hip — the LayerNorm-fused instantiations of gemm_kernel (gemm_kernel.h) on the row-complete 64 x 256 tile (round 6): d_model = 256
// between 17 500 and ~37 000 frames, where 128-row tiles number fewer than the chip's workgroup slots (B = 64 x 500: 250 tiles on 256
// CUs, one workgroup per CU and nothing beside its epilogue).  A translation unit of its own so that it compiles next to the others.
#include "gemm_kernel.h"

namespace smx {

template <bool B_KC>
static int launch_ln_r64(GemmParams& p, hipStream_t s) {
  typedef bf16_t T;
  p.tiles_n = (p.N + 63) / 64;
  p.tiles_m = 1;
  const dim3 grid(p.tiles_n), block(256);
  const bool lnb = (p.e.flags & SMX_EPI_LN_BWD) != 0, lnf = (p.e.flags & SMX_EPI_LN_FWD) != 0, xf32 = (p.e.io_flags & SMX_IO_LNX_F32) != 0;
  const bool ext = lnb && (p.e.lnf_act != SMX_ACT_NONE || p.e.z);
  if (plan_only(p, 0, true, B_KC, 64, 256, true, ext ? (xf32 ? 7 : 3) : (lnb ? (xf32 ? 5 : 1) : (lnf ? 2 : 0)), 0)) return SMX_OK;
  if (ext && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 64, 256, true, 7>), grid, block, 0, s, p);
  else if (ext) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 64, 256, true, 3>), grid, block, 0, s, p);
  else if (lnb && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 64, 256, true, 5>), grid, block, 0, s, p);
  else if (lnb) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 64, 256, true, 1>), grid, block, 0, s, p);
  else if (lnf) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 64, 256, true, 2>), grid, block, 0, s, p);
  else return fail(SMX_EINVAL, "launch_ln_fused_256_r64: no LayerNorm flag");
  return check_launch("smx_gemm");
}

int launch_ln_fused_256_r64(GemmParams& p, bool b_kc, hipStream_t s) { return b_kc ? launch_ln_r64<true>(p, s) : launch_ln_r64<false>(p, s); }

}  // namespace smx
