// gemm_ln256.hip — the LayerNorm-fused instantiations of gemm_kernel (gemm_kernel.h) on the row-complete 128 x 256 tile (two workgroups per CU; d_model = 256).
// A translation unit of its own so that it compiles next to gemm.hip (each of these kernels keeps 256-512 registers per lane
// and takes hipcc 10-20 s).
#include "gemm_kernel.h"

namespace smx {

int launch_ln_fused_256_r64(GemmParams& p, bool b_kc, hipStream_t s);      // gemm_ln256r64.hip: the 64 x 256 tile

template <bool B_KC>
static int launch_ln(GemmParams& p, hipStream_t s) {
  typedef bf16_t T;
  if (ln_tile_rows_for(p.N, 256) == 64) return launch_ln_fused_256_r64(p, B_KC, s);
  p.tiles_n = (p.N + 127) / 128;
  p.tiles_m = 1;
  const dim3 grid(p.tiles_n), block(256);
  const bool lnb = (p.e.flags & SMX_EPI_LN_BWD) != 0, lnf = (p.e.flags & SMX_EPI_LN_FWD) != 0, xf32 = (p.e.io_flags & SMX_IO_LNX_F32) != 0;
  const bool ext = lnb && (p.e.lnf_act != SMX_ACT_NONE || p.e.z);
  if (plan_only(p, 0, true, B_KC, 128, 256, true, ext ? (xf32 ? 7 : 3) : (lnb ? (xf32 ? 5 : 1) : (lnf ? 2 : 0)), 0)) return SMX_OK;
  if (ext && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 256, true, 7>), grid, block, 0, s, p);
  else if (ext) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 256, true, 3>), grid, block, 0, s, p);
  else if (lnb && xf32) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 256, true, 5>), grid, block, 0, s, p);
  else if (lnb) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 256, true, 1>), grid, block, 0, s, p);
  else if (lnf) hipLaunchKernelGGL((gemm_kernel<T, true, B_KC, 128, 256, true, 2>), grid, block, 0, s, p);
  else return fail(SMX_EINVAL, "launch_ln_fused_256: no LayerNorm flag");
  return check_launch("smx_gemm");
}

// p: a bf16 NT (b_kc) / NN GEMM with M == 256, whole aligned vectors, splits == 1, batch == 1 (checked by launch_layout, gemm.hip)
int launch_ln_fused_256(GemmParams& p, bool b_kc, hipStream_t s) { return b_kc ? launch_ln<true>(p, s) : launch_ln<false>(p, s); }

}  // namespace smx
