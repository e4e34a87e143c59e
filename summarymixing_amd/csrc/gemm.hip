// gemm.hip — batched strided MFMA GEMM with the fused SummaryMixing epilogue, gfx950 only.
//
//   C[b] (N x M) = epilogue( op(A[b]) . op(B[b]) )      fp32 accumulate
//
// Design (MI355X first):
//  * 256-thread workgroups = 4 wave64 in a 2x2 arrangement; wave tile = (TILE_N/2) x (TILE_M/2) built from
//    32x32 MFMA fragments: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32).
//  * the MFMA is issued with SWAPPED operands (first operand = m side, second = n side) so that a lane owns
//    4 CONSECUTIVE output columns of ONE output row per accumulator quad: the epilogue reads bias / C0 /
//    residual and writes Y/Z with 8-byte (bf16) or 16-byte (fp32) vector accesses instead of 2-byte scatters.
//  * operands are staged HBM -> registers -> LDS (issue-early / write-late, one LDS buffer, 2 barriers per
//    K tile); the next tile's global loads are in flight while the current tile is multiplied.
//  * LDS images:   bf16, reduce-contiguous ("KC"): [row][64 k], 16-byte chunks XOR-swizzled by (row>>1)&7
//                        -> conflict-free ds_read_b128 fragment reads;
//                  bf16, reduce-strided ("KS"):    [k/4][row][4] (a 4x8 register transpose on the way in),
//                        16-byte granules XOR-swizzled inside each 32-row window -> conflict-free ds_read_b64
//                        reads AND conflict-free ds_write_b128 stores;
//                  fp32: [k][row] for both kinds (fragment = one dword per lane, rows consecutive).
//    so NT (forward), NN (dgrad / per-head einsum) and TN (wgrad) all feed the same MFMA loop without any
//    transposed copy of activations or weights in HBM.
//  * blockIdx is remapped so that the M-tiles that share one A row panel land on the same XCD (same L2).
//  * TN/wgrad reduces over the (long) frame dimension: split-K over blockIdx.y with fp32 atomics into the
//    caller-zeroed gradient buffer.
#include <utility>

#include "gemm_kernel.h"

namespace smx {



// ---- wgrad (TN) with both operands on the LDS-DMA path ---------------------------------------------------------------
// dW = dZ^T X reduces over the frames: both operands are reduce-strided (a k row = 128 contiguous columns = 256 B), the
// loop is long (rows / splits / 64 steps) and has no epilogue work inside - the shape the LDS-DMA ring fits best.  Per
// K step the register-staged kernel spends ~850 cycles writing the stage to LDS and on its two barriers, ~950 waiting for
// the loads and ~800 in MFMAs, one after the other (tools/gemm_stamps.py: 2.6 K cycles per step at two waves per SIMD).
// Here a stage (64 k rows x 128 columns of dZ and of X, 32 KB) goes HBM/L2 -> LDS by global_load_lds_dwordx4, two stages
// form a ring, ONE barrier per K step, no operand ever touches a VGPR (two workgroups per CU as before).
// LDS image of an operand stage: element (k, c) at  k * 256 + (((c >> 3) ^ ((k & 3) << 2)) << 4) + (c & 7) * 2.
// The DMA writes a piece linearly (lane i -> +16 i; a 1 KB piece = 4 k rows), so the XOR is applied to the SOURCE column
// granule; it replaces the +64 B row pad of the register-staged image.  A 256-byte k row covers all 64 LDS banks once, so
// the bank of an access is its granule index: the 32 lanes served together by ds_read_b64_tr_b16 touch 4 k rows x 4
// granules, and XOR-ing the granule with 4 * (k & 3) sends the four rows to four disjoint granule quadruples (with
// 2 * (k & 3) rows 0/1 and 2/3 collided: SQ_LDS_BANK_CONFLICT = 2 cycles per read).
// Bias gradient (column sums of dZ): one extra MFMA per fragment against a constant all-ones B fragment in the waves
// that own output columns 0..63 of the first column tile - no LDS reads, no VALU.

constexpr int kTnBk = 64;      // k rows per LDS-DMA stage of the wgrad kernel (64: ring of 2; 32: ring of 4 - measured 3-5 % slower)
__global__ __launch_bounds__(256, 2) void gemm_tn_dma_kernel(GemmParams p) {
  typedef bf16_t T;
  // ring of NST thin stages (BK k rows each): the DMA of stage it + NST - 1 is issued while stage it is multiplied, i.e.
  // a prefetch distance of (NST - 1) * BK frames in the same 64 KB of LDS
  constexpr int BK = kTnBk, NST = 128 / BK, TILE = 128, WN = 64, WM = 64, FN = 2, FM = 2;
  constexpr int OP_BYTES = BK * TILE * 2, STAGE_BYTES = 2 * OP_BYTES;
  constexpr int NPC = BK / 16;                          // 1 KB pieces per wave, operand and stage
  static_assert(BK == 64 || BK == 32, "ring of 2 x 64 or 4 x 32 k rows");
  constexpr int PH_ROWS = 64, NPH = 2, STG_LD = TILE * 4 + 16, EPI_BYTES = (PH_ROWS * STG_LD + 63) / 64 * 64;
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE_BYTES];      // the ring; the epilogue rows alias it
  float* side = reinterpret_cast<float*>(smem + EPI_BYTES);                   // (written after the main loop)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wm = wave & 1, l31 = lane & 31, hi = lane >> 5;
  int tile_n, tile_m, bz, split;
  {
    const int ntiles = p.tiles_n * p.tiles_m;
    if (p.splits == 1) {
      int bid = blockIdx.x;
      const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      tile_n = bid / p.tiles_m; tile_m = bid % p.tiles_m;
      bz = blockIdx.y; split = 0;
    } else {                                             // a whole split lives on one XCD (see gemm_kernel)
      const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
      const int per = ntiles * p.batch;
      const int item = idx % per;
      split = (idx / per) * 8 + xcd;
      if (split >= p.splits) return;
      bz = item / ntiles;
      const int tl = item % ntiles;
      tile_n = tl / p.tiles_m; tile_m = tl % p.tiles_m;
    }
  }
  const int n0 = tile_n * TILE, m0 = tile_m * TILE;
  const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int niter = (kend - kbeg) / BK;
  const T* A = reinterpret_cast<const T*>(p.A) + (long)bz * p.sA;
  const T* B = reinterpret_cast<const T*>(p.B) + (long)bz * p.sB;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const int prow = lane >> 4, gsrc = ((lane & 15) ^ (prow << 2)) * 8;      // this lane's k row in a piece, source column
  // source pointers of this lane's 4 + 4 pieces, advanced by one K step per issue (no per-step address arithmetic
  // beyond eight 64-bit adds)
  const T* pa[NPC];
  const T* pb[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {                        // BK / 4 pieces of 4 k rows per operand and stage, NPC per wave
    const long kr = kbeg + 4 * (wave + 4 * j) + prow;
    pa[j] = A + kr * p.lda + n0 + gsrc;
    pb[j] = B + kr * p.ldb + m0 + gsrc;
  }
  const long stepa = (long)BK * p.lda, stepb = (long)BK * p.ldb;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  auto issue = [&](int it) {
    const uint32_t dst = wave_lds + (it % NST) * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      glds16(pa[j], dst + j * 4096);
      glds16(pb[j], dst + OP_BYTES + j * 4096);
      pa[j] += stepa;
      pb[j] += stepb;
    }
  };
  f32x16 acc[FN][FM], accb[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) accb[i][q] = 0.f;
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  }
  const bool do_cs = p.acolsum != nullptr && tile_m == 0 && wm == 0;      // (uniform per wave)
  const uint32_t one2 = 0x3F803F80u;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(one2, one2, one2, one2));
  for (int s_ = 0; s_ < NST - 1 && s_ < niter; ++s_) issue(s_);
  for (int it = 0; it < niter; ++it) {
    // this wave's pieces of stage `it` have landed when at most the (2 NPC each) DMA instructions of the younger stages in
    // flight are outstanding (vmcnt retires in order; nothing else uses vector memory in this loop)
    const int ahead = min(NST - 2, niter - 1 - it);
    if (ahead >= 2 && NST >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPC) : "memory");
    else if (ahead >= 1 && NST >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                       // ... and everybody's; the stage read last step is free again
    if (it + NST - 1 < niter) issue(it + NST - 1);
    const char* As = smem + (it % NST) * STAGE_BYTES;
    const char* Bs = As + OP_BYTES;
    // fragments double-buffered in registers: the transposing LDS reads of sub-step kk + 1 are in flight under the
    // MFMAs of sub-step kk
    bf16x8 fa[2][FN], fb[2][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i) fa[0][i] = frag_tr_swz(As, wn * WN + i * 32 + l31, 0, hi);
#pragma unroll
    for (int j = 0; j < FM; ++j) fb[0][j] = frag_tr_swz(Bs, wm * WM + j * 32 + l31, 0, hi);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < FN; ++i) fa[nxt][i] = frag_tr_swz(As, wn * WN + i * 32 + l31, kk + 1, hi);
#pragma unroll
        for (int j = 0; j < FM; ++j) fb[nxt][j] = frag_tr_swz(Bs, wm * WM + j * 32 + l31, kk + 1, hi);
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < FN; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fa[cur][i], accb[i], 0, 0, 0);
      }
    }
  }
  if (do_cs && hi == 0) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WN + i * 32 + l31;
      if (n < p.N) p.acolsum[((long)split * p.batch + bz) * p.N + n] = accb[i][0];
    }
  }
  // ---- epilogue (fp32 slabs or the final output) through the staged rows, as in gemm_kernel -------------------------
  const smx_epilogue& e = p.e;
  lds_barrier();                                         // every wave is done reading the ring
  if (t < TILE) side[t] = (e.bias && m0 + t < p.M) ? e.bias[(long)bz * e.bias_batch_stride + m0 + t] : 0.f;
  else side[t] = ((e.row_mask && n0 + t - TILE < p.N) ? (e.row_mask[n0 + t - TILE] ? 1.f : 0.f) : 1.f) * e.alpha;
  const int osz = (e.out_mode == SMX_OUT_T) ? 2 : 4;
#pragma unroll 1
  for (int ph = 0; ph < NPH; ++ph) {
    lds_barrier();
    if (wn == ph) {
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(smem + (i * 32 + l31) * STG_LD + (wm * WM + j * 32 + g * 8 + hi * 4) * 4) =
                make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
    }
    lds_barrier();
    // (the dispatch admits no element-wise side input here: the SIMPLE instantiation)
    if (osz == 2) epilogue_phase<T, 2, TILE, TILE, true, 1>(p, smem, side, ph, n0 + ph * PH_ROWS, m0, bz, split, t);
    else epilogue_phase<T, 4, TILE, TILE, true, 1>(p, smem, side, ph, n0 + ph * PH_ROWS, m0, bz, split, t);
  }
}

static int launch_tn_dma(GemmParams& p, hipStream_t s) {
  p.tiles_n = p.N / 128;
  p.tiles_m = p.M / 128;
  dim3 grid(p.tiles_n * p.tiles_m, p.batch);
  if (p.splits > 1) grid = dim3(8 * p.tiles_n * p.tiles_m * p.batch * ((p.splits + 7) / 8), 1);
  if (plan_only(p, 1, false, false, 128, 128, true, 0, 0)) return SMX_OK;
  hipLaunchKernelGGL(gemm_tn_dma_kernel, grid, dim3(256), 0, s, p);
  return check_launch("smx_gemm");
}

// ---- host dispatch ------------------------------------------------------------------------------------------
template <typename T, bool A_KC, bool B_KC, int TN, int TM>
static int launch_tile(GemmParams& p, bool vec, hipStream_t s) {
  p.tiles_n = (p.N + TN - 1) / TN;
  p.tiles_m = (p.M + TM - 1) / TM;
  dim3 grid(p.tiles_n * p.tiles_m, p.batch);
  if (p.splits > 1) grid = dim3(8 * p.tiles_n * p.tiles_m * p.batch * ((p.splits + 7) / 8), 1);
  if (plan_only(p, 0, A_KC, B_KC, TN, TM, vec, 0, 0)) return SMX_OK;
  if (vec) hipLaunchKernelGGL((gemm_kernel<T, A_KC, B_KC, TN, TM, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<T, A_KC, B_KC, TN, TM, false>), grid, dim3(256), 0, s, p);
  if (p.e.colsum) launch_colsum_partials(reinterpret_cast<const float*>(p.e.workspace), p.tiles_n, p.M, p.e.colsum, s);
  return check_launch("smx_gemm");
}

template <typename T, bool A_KC, bool B_KC>
static int launch_layout(GemmParams& p, bool vec, hipStream_t s) {
  if (p.e.flags & (SMX_EPI_LN_BWD | SMX_EPI_LN_FWD)) {
    // fused LayerNorm: the tile must hold whole rows -> the 128 x 256 tile (two workgroups per CU) or the 128 x 512 tile (one),
    // whatever the grid size
    if constexpr (sizeof(T) == 2 && A_KC) {
      const bool lnb = (p.e.flags & SMX_EPI_LN_BWD) != 0;
      // (the forward variant may write the new fp32 stream tensor: SMX_OUT_F32; the backward variants emit dtype T)
      const bool ln_ok = vec && p.N >= 128 && p.splits == 1 && p.batch == 1 && !p.e.colsum &&
                         (p.e.out_mode == SMX_OUT_T || (!lnb && p.e.out_mode == SMX_OUT_F32));
      // (the instantiations live in gemm_ln256.hip / gemm_ln512.hip: their own translation units, compiled in parallel)
      if (ln_ok && p.M == 512) return launch_ln_fused_512(p, B_KC, s);
      if (ln_ok && p.M == 256) return launch_ln_fused_256(p, B_KC, s);
    }
    return fail(SMX_EUNSUPPORTED, "smx_gemm: fused LayerNorm needs bf16 NT / NN, M == 256 or 512, N >= 128 and 16-byte aligned operands");
  }

  // big tiles once they alone fill the chip (256 CUs x 2 resident blocks); otherwise 64x64 for more blocks
  long big = (long)((p.N + 127) / 128) * ((p.M + 127) / 128) * p.batch * p.splits;
  constexpr int force_small = 0;
  // wide 128 x 256 tile (2 workgroups per CU, 128 accumulator registers per lane): when it covers the whole output
  // width (M == 256: the activation panel is fetched exactly once and 500 tiles fill the 512 slots in one round at
  // 64000 frames) or when the reduction is long enough for the doubled MFMA-per-LDS-read ratio to matter.
  // Measured at 64000 frames: (K=1024, M=256) NT 77 -> 59 us, NN 65 -> 55 us; (K=256, M=1024) 99 -> 103 us (not used).
  // Round 4, d_model = 512 (M = 512: two column tiles per row panel, so the panel is fetched twice either way): K = 512 .. 1536 on
  // the 128 x 128 tile instead - twice the workgroups per launch - is C5 forward 62.64 -> 61.87 ms, C2a 51.4 -> 51.1, C4 40.9 -> 40.75
  // (SMX_GEMM_WIDE=0 / 1024 / 2048 A/B on one box, twice); K = 2048 is a tie and stays wide.  SMX_GEMM_WIDE=<k >= 2>: that minimum K.
  // (Also measured and NOT taken: the wide tile for the up-projection shapes M >= 1024 - C5 +0.3 ms, C2a +1.1, C2b +0.7 -, the pipelined
  //  main loop on the 128 x 128 tile, -DSMX_WIDE_PIPE=2 - C2a +0.6, C5 +2.4, C4 +1.0 -, two register stages, -DSMX_NS_KC=2 - +0.1..0.3.)
  const bool wide = p.M == 256 || (p.K >= 2048 && p.M <= 512);
  // wgrad-shaped TN GEMMs with both operands on the LDS-DMA ring (gemm_tn_dma_kernel)
  if constexpr (sizeof(T) == 2 && !A_KC && !B_KC) {
    if (vec && p.N % 128 == 0 && p.M % 128 == 0 && p.K % 64 == 0 && p.kchunk % 64 == 0 && p.K >= 64 &&
        p.e.out_mode != SMX_OUT_ATOMIC_F32 && !p.e.colsum && !p.e.res && !p.e.c0 && !p.e.z && !p.ablate &&
        (long)(p.N / 128) * (p.M / 128) * p.batch * p.splits >= 256)
      return launch_tn_dma(p, s);
  }
  // 256 x 256 tile, one workgroup of four waves per CU (128 x 128 per wave, 256 accumulator registers per lane): half the
  // LDS fragment reads per MFMA and two thirds of the operand staging of two 128 x 256 workgroups - the library's tile for the
  // long reductions (SMX_T256=1 while it is being measured)
  if constexpr (sizeof(T) == 2 && A_KC) {
    // Measured at 64 000 frames against the 128 x 256 tile (tools/experiments/ab_t256.sh): K = 2048 -> 512 dgrad 198 -> 169 us,
    // forward + bias + dropout + residual 204 -> 195 us, 240 000 x 2048 -> 512 bias only 667 -> 623 us; K = 1024 ties or loses.
    // With ONE workgroup per CU nothing runs beside the epilogue, so a heavy one eats the gain: in the steps the fp32-stream
    // down-projection (reads and writes 4-byte rows) got SLOWER (config 5 forward 61.5 -> 63.0 ms), the plain dgrad faster.
    // Hence: K >= 2048 and an epilogue without element-wise side inputs (epi_simple == 1) writing dtype T.
    // SMX_T256=0 off, 2 = every eligible shape (tests).
    const int t256 = cfg().t256;
    // SMX_T256=3 (experiments / tests): M == 512 shapes on the row-complete 128 x 512 tile of the LayerNorm-fused kernels
    if (t256 == 3 && vec && p.N >= 128 && p.M == 512 && p.splits == 1 && p.batch == 1 && p.K >= 128) return launch_ln_fused_512(p, B_KC, s);
    if (t256 && vec && !force_small && p.N >= 256 && p.M % 256 == 0 && p.splits == 1 && p.batch == 1 &&
        (t256 >= 2 || (p.K >= 2048 && p.epi_simple == 1 && p.e.out_mode == SMX_OUT_T && !p.e.z)) && p.K >= 128 &&
        (long)((p.N + 255) / 256) * (p.M / 256) >= (t256 >= 2 ? 1 : 200)) {
      p.tiles_n = (p.N + 255) / 256;
      p.tiles_m = p.M / 256;
      if (plan_only(p, 0, A_KC, B_KC, 256, 256, true, 0, 0)) return SMX_OK;
      hipLaunchKernelGGL((gemm_kernel<T, A_KC, B_KC, 256, 256, true>), dim3(p.tiles_n * p.tiles_m, 1), dim3(256), 0, s, p);
      if (p.e.colsum) launch_colsum_partials(reinterpret_cast<const float*>(p.e.workspace), p.tiles_n, p.M, p.e.colsum, s);
      return check_launch("smx_gemm");
    }
  }
#ifndef SMX_WIDE_MIN_TILES
#define SMX_WIDE_MIN_TILES 256
#endif
  if (wide && !force_small && p.N >= 128 && p.M >= 256 && p.M % 256 == 0 && p.splits == 1 &&
      (long)((p.N + 127) / 128) * (p.M / 256) * p.batch >= SMX_WIDE_MIN_TILES)
    return launch_tile<T, A_KC, B_KC, 128, 256>(p, vec, s);
  if (!force_small && big >= 256 && p.N >= 128 && p.M >= 128) return launch_tile<T, A_KC, B_KC, 128, 128>(p, vec, s);
  return launch_tile<T, A_KC, B_KC, 64, 64>(p, vec, s);
}

template <typename T>
static int launch_dtype(int layout, GemmParams& p, bool vec, hipStream_t s) {
  switch (layout) {
    case SMX_GEMM_NT: return launch_layout<T, true, true>(p, vec, s);
    case SMX_GEMM_NN: return launch_layout<T, true, false>(p, vec, s);
    case SMX_GEMM_TN: return launch_layout<T, false, false>(p, vec, s);
  }
  return fail(SMX_EINVAL, "smx_gemm: unknown layout %d", layout);
}

}  // namespace smx

using namespace smx;

#ifdef SMX_DIAG   // diagnostic build only (libsmx_diag.so): the product library has no global mutable state and no debug export
long long* g_dbg_stamps = nullptr;   // (tools/gemm_stamps.py; the parked tools/experiments/ffn_fused kernel reads it too)
extern "C" void smx_debug_set_timing_buffer(void* p) { g_dbg_stamps = reinterpret_cast<long long*>(p); }
#endif

static int gemm_impl(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B,
                     int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K,
                     int batch, int splits, int64_t split_stride, const smx_epilogue* epi, void* stream,
                     float* acolsum = nullptr, const int* conv = nullptr, smx_gemm_plan* plan = nullptr) {
  // conv = {T, F}: the K-contiguous operand A (NT) / the reduce-strided operand B (TN) is the implicit 3x3-stride-2 patch matrix
  // of the channels-last tensor (batches, T, F, 64) behind that pointer (bf16, 64 channels: gemm_kernel<..., GATHER>)
  SMX_REQUIRE(A && B && C, "smx_gemm: null operand");
  SMX_REQUIRE(N >= 0 && M >= 0 && K >= 0 && batch >= 1 && splits >= 1, "smx_gemm: bad sizes N=%d M=%d K=%d", N, M, K);
  SMX_REQUIRE(dtype == SMX_F32 || dtype == SMX_BF16, "smx_gemm: bad dtype %d", dtype);
  if (N == 0 || M == 0) return SMX_OK;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.N = N; p.M = M; p.K = K; p.batch = batch;
  if (epi) p.e = *epi;
  else p.e.alpha = 1.f;
  const int BK = dtype == SMX_BF16 ? 64 : 32;
  if (splits > 1) {
    SMX_REQUIRE(p.e.out_mode == SMX_OUT_ATOMIC_F32 || (p.e.out_mode == SMX_OUT_F32 && split_stride > 0),
                "smx_gemm: splits>1 needs SMX_OUT_ATOMIC_F32, or SMX_OUT_F32 slabs (smx_linear_wgrad)");
    int kc = (K + splits - 1) / splits;
    kc = ((kc + BK - 1) / BK) * BK;
    p.kchunk = kc;
    splits = (K + kc - 1) / kc;
    if (splits < 1) splits = 1;
  } else {
    p.kchunk = K > 0 ? ((K + BK - 1) / BK) * BK : BK;
  }
  p.splits = splits;
  if (p.e.out_mode == SMX_OUT_ATOMIC_F32)
    SMX_REQUIRE(!p.e.bias && !p.e.c0 && !p.e.z && !p.e.res && !p.e.row_mask && p.e.act == SMX_ACT_NONE && p.e.drop_p == 0.f,
                "smx_gemm: atomic output takes no epilogue");
  SMX_REQUIRE(p.e.drop_p >= 0.f && p.e.drop_p < 1.f, "smx_gemm: 0 <= drop_p < 1");
  SMX_REQUIRE(p.e.c0_mode == SMX_C0_NONE || (p.e.c0 && (p.e.c0_mode == SMX_C0_ROW || p.e.c0_div > 0)),
              "smx_gemm: bad C0 spec");
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  const int vpt = dtype == SMX_BF16 ? 8 : 4;
  // vector (16-byte) operand loads need aligned bases / strides and whole vectors along the contiguous dim
  bool a_kc = layout != SMX_GEMM_TN, b_kc = layout == SMX_GEMM_NT;
  bool vec = aligned16(A) && aligned16(B) && lda % vpt == 0 && ldb % vpt == 0 && (strideA * es) % 16 == 0 &&
             (strideB * es) % 16 == 0;
  vec = vec && (a_kc ? K % vpt == 0 : N % vpt == 0) && (b_kc ? K % vpt == 0 : M % vpt == 0);
  // bf16 vector kernels fetch their operand stages with buffer loads (BufStage): a reduce-contiguous operand then needs
  // whole 64-element K steps, and every operand must span less than 2 GB
  if (dtype == SMX_BF16 && vec) {
    const long span_a = a_kc ? (long)N * lda : (long)K * lda, span_b = b_kc ? (long)M * ldb : (long)K * ldb;
    if (((a_kc || b_kc) && K % 64 != 0) || span_a * 2 >= (1L << 31) || span_b * 2 >= (1L << 31)) vec = false;
  }
  // 4-wide epilogue accesses
  auto ok4 = [&](const void* ptr, int64_t ld, size_t esz) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) % (4 * esz)) == 0 && ld % 4 == 0);
  };
  size_t cs = p.e.out_mode == SMX_OUT_T ? es : 4;
  p.epi_vec = ok4(C, ldc, cs) && ok4(p.e.z, p.e.ldz, es) && ok4(p.e.res, p.e.ldr, es) && ok4(p.e.bias, 4, 4) &&
              ok4(p.e.c0, p.e.ldc0, 4) && (strideC % 4 == 0) && (p.e.bias_batch_stride % 4 == 0);
  {
    // LDS-staged coalesced stores need 16-byte addressable output rows
    auto ok16 = [&](const void* ptr, int64_t ld, int64_t bs, size_t esz) {
      return ptr == nullptr || (aligned16(ptr) && (ld * (int64_t)esz) % 16 == 0 && (bs * (int64_t)esz) % 16 == 0);
    };
    p.epi_lds = p.e.out_mode != SMX_OUT_ATOMIC_F32 && ok16(C, ldc, strideC, cs) && ok16(p.e.z, p.e.ldz, strideC, es) &&
                ok16(p.e.res, p.e.ldr, strideC, es) && ok16(p.e.bias, 4, p.e.bias_batch_stride, 4) &&
                ok16(p.e.c0, p.e.ldc0, 0, 4) && (split_stride * (int64_t)cs) % 16 == 0 && M % 4 == 0;
    p.sSplit = split_stride;
  }
  // the vector kernels take whole 16-byte items in the epilogue as well (no per-element guards anywhere)
  vec = vec && p.epi_lds && M % (int)(16 / cs) == 0;
  if (p.e.flags & SMX_EPI_ACT_GRAD)
    SMX_REQUIRE(p.e.z && !p.e.res && batch == 1 && splits == 1 && p.e.out_mode != SMX_OUT_ATOMIC_F32,
                "smx_gemm: SMX_EPI_ACT_GRAD needs z (input), no residual, batch == 1, splits == 1");
  if (p.e.io_flags & SMX_IO_RES_F32)
    SMX_REQUIRE(p.e.res && p.e.out_mode == SMX_OUT_F32 && !(p.e.flags & (SMX_EPI_ACT_GRAD | SMX_EPI_LN_BWD)) && aligned16(p.e.res) && p.e.ldr % 4 == 0,
                "smx_gemm: SMX_IO_RES_F32 needs a 16-byte aligned float32 `res`, out_mode SMX_OUT_F32 and no ACT_GRAD / LN_BWD");
  if (p.e.io_flags & SMX_IO_LNX_F32)
    SMX_REQUIRE((p.e.flags & SMX_EPI_LN_BWD) && p.e.ln_ldx % 4 == 0, "smx_gemm: SMX_IO_LNX_F32 goes with SMX_EPI_LN_BWD");
  if (p.e.flags & SMX_EPI_LN_BWD)
    SMX_REQUIRE(p.e.ln_x && p.e.ln_stats && p.e.ln_gamma && p.e.ln_partial && !p.e.bias && !p.e.c0 && !p.e.row_mask &&
                (p.e.z ? (p.e.ln_dx2 && aligned16(p.e.z) && p.e.ldz % 8 == 0) : p.e.act == SMX_ACT_NONE) &&
                (p.e.lnf_act == SMX_ACT_NONE || p.e.lnf_beta) && p.e.drop_p == 0.f && p.e.alpha == 1.f && aligned16(p.e.ln_x) && p.e.ln_ldx % 8 == 0 &&
                (!p.e.ln_dx2 || (aligned16(p.e.ln_dx2) && p.e.ln_lddx2 % 8 == 0)) && p.e.ln_drop_p2 >= 0.f && p.e.ln_drop_p2 < 1.f,
                "smx_gemm: SMX_EPI_LN_BWD takes ln_x / ln_stats / ln_gamma / ln_partial (+ res, ln_dx2) and no other epilogue field");
  if (p.e.io_flags & SMX_IO_LNFY_F32) SMX_REQUIRE(p.e.flags & SMX_EPI_LN_FWD, "smx_gemm: SMX_IO_LNFY_F32 goes with SMX_EPI_LN_FWD");
  if (p.e.flags & SMX_EPI_LN_FWD)
    SMX_REQUIRE(p.e.lnf_gamma && p.e.lnf_beta && p.e.lnf_y && aligned16(p.e.lnf_y) && p.e.lnf_ldy % 8 == 0 &&
                !(p.e.flags & (SMX_EPI_LN_BWD | SMX_EPI_ACT_GRAD)), "smx_gemm: SMX_EPI_LN_FWD needs lnf_gamma / lnf_beta / lnf_y");
  if (p.e.lnf2_y) {
    SMX_REQUIRE((p.e.flags & SMX_EPI_LN_FWD) && p.e.lnf2_gamma && p.e.lnf2_beta && aligned16(p.e.lnf2_y) && p.e.lnf2_ldy % 8 == 0,
                "smx_gemm: lnf2_* (second LayerNorm) goes with SMX_EPI_LN_FWD and needs gamma / beta / an aligned output");
    if (!(smx_gemm_ln_pair_ok(dtype, N, M, K) && p.e.out_mode == SMX_OUT_F32 && (p.e.io_flags & SMX_IO_RES_F32) && p.e.res && !p.e.c0 &&
          !p.e.colsum && batch == 1 && splits == 1))
      return fail(SMX_EUNSUPPORTED, "smx_gemm: the second LayerNorm (lnf2_*) needs a row-complete tile and the float32-stream epilogue (smx_gemm_ln_pair_ok)");
  }
  if (p.e.colsum)
    SMX_REQUIRE(p.e.workspace && batch == 1 && splits == 1 && p.e.out_mode != SMX_OUT_ATOMIC_F32,
                "smx_gemm: colsum needs a workspace (smx_gemm_colsum_workspace), batch == 1, splits == 1");
  // Store policy.  Z (the pre-activation saved for the backward pass) is not read again for a long time: always
  // streamed past the caches.  The output C is consumed by the next kernel: streamed only when it is too large to
  // survive in the 256 MB MALL anyway (measured: FFN up-projection at 64000 frames 118 -> 96 us, no change at 32000).
  constexpr long nt_bytes = 96L << 20;                      // (re-swept in round 4: 50 / 140 / 280 MB lose 0.1-0.9 ms per step)
  p.nt = 1 | (((long)N * M * (long)cs * batch >= nt_bytes) ? 2 : 0);
  // With a LayerNorm appended (SMX_EPI_LN_FWD) the next kernel reads the LayerNorm output, not C: C (the float32 stream tensor, or
  // the pre-norm tensor the backward pass wants) is streamed whatever its size, so that it does not push the LayerNorm output out
  // of the Infinity Cache (the same finding as smx_layernorm_fwd_pair_x32, rowwise.hip).
  if (p.e.flags & SMX_EPI_LN_FWD) p.nt |= 2;
  // epilogue instantiation: 1 = no element-wise side input, 2 = one (residual / saved pre-activation), 0 = general (C0 rows, column sums)
  p.epi_simple = (p.e.c0 || p.e.colsum) ? 0 : ((p.e.res || (p.e.flags & SMX_EPI_ACT_GRAD)) ? 2 : 1);
  p.ablate = cfg().gemm_ablate;                          // (0 unless built with -DSMX_DIAG)
#ifdef SMX_DIAG
  p.dbg = g_dbg_stamps;
#endif
  p.epoch = p.e.epoch;
  p.acolsum = acolsum;
  p.plan = plan;
  p.dthresh = (unsigned)((double)p.e.drop_p * 4294967296.0);
  SMX_REQUIRE(p.e.drop_cols >= 0 && p.e.drop_cols <= M && p.e.drop_cols % 8 == 0, "smx_gemm: drop_cols must be a multiple of 8 in [0, M]");
  p.drop_cols = p.e.drop_cols > 0 ? p.e.drop_cols : M;
  p.dscale = 1.f / (1.f - p.e.drop_p);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (conv && conv[0] < 0) {
    // folded DFT frames: conv = {-3 | -4, n_fft}; A = zero-padded waveform rows (lda = hop), float32 NT, batch = utterances
    SMX_REQUIRE(dtype == SMX_F32 && layout == SMX_GEMM_NT && splits == 1 && conv[1] % 8 == 0 &&
                    K == (conv[0] == -3 ? conv[1] / 2 + 4 : conv[1] / 2) && lda % 4 == 0 && aligned16(A),
                "smx_gemm (folded DFT operand): float32 NT, K = n_fft / 2 (+ 4 for the cosine part)");
    p.g_T = conv[1];
    p.gather = -conv[0];
    p.tiles_n = (p.N + 127) / 128;
    p.tiles_m = (p.M + 127) / 128;
    dim3 grid(p.tiles_n * p.tiles_m, p.batch);
    if (plan_only(p, 0, true, true, 128, 128, true, 0, conv[0] == -3 ? 3 : 4)) return SMX_OK;
    if (conv[0] == -3) hipLaunchKernelGGL((gemm_kernel<float, true, true, 128, 128, true, 0, 3>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<float, true, true, 128, 128, true, 0, 4>), grid, dim3(256), 0, s, p);
    return check_launch("smx_gemm (folded DFT operand)");
  }
  if (conv) {
    const long rows = layout == SMX_GEMM_NT ? N : K;      // patch rows = output pixels
    p.g_T = conv[0]; p.g_F = conv[1]; p.g_T2 = (conv[0] + 1) / 2; p.g_F2 = (conv[1] + 1) / 2;
    const long nb = rows / ((long)p.g_T2 * p.g_F2);
    p.g_npix = nb * p.g_T * p.g_F;
    SMX_REQUIRE(dtype == SMX_BF16 && vec && batch == 1 && rows == nb * p.g_T2 * p.g_F2 && p.g_npix * 128 < (1L << 31) &&
                    (layout == SMX_GEMM_NT ? (K == 576 && splits == 1) : (layout == SMX_GEMM_TN && M == 576)),
                "smx_gemm (implicit conv operand): bf16, 64 channels, whole batches of patch rows, input < 2 GB");
    p.gather = layout == SMX_GEMM_NT ? 1 : 2;
    p.tiles_n = (p.N + 63) / 64;
    p.tiles_m = (p.M + 63) / 64;
    dim3 grid(p.tiles_n * p.tiles_m, 1);
    if (p.splits > 1) grid = dim3(8 * p.tiles_n * p.tiles_m * ((p.splits + 7) / 8), 1);
    if (plan_only(p, 0, layout == SMX_GEMM_NT, layout == SMX_GEMM_NT, 64, 64, true, 0, layout == SMX_GEMM_NT ? 1 : 2)) return SMX_OK;
    if (layout == SMX_GEMM_NT) hipLaunchKernelGGL((gemm_kernel<bf16_t, true, true, 64, 64, true, 0, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<bf16_t, false, false, 64, 64, true, 0, 2>), grid, dim3(256), 0, s, p);
    return check_launch("smx_gemm (implicit conv operand)");
  }
  if (dtype == SMX_BF16) return launch_dtype<bf16_t>(layout, p, vec, s);
  return launch_dtype<float>(layout, p, vec, s);
}

extern "C" int smx_gemm_ln_fused_ok(int dtype, int N, int M, int K) {
  return dtype == SMX_BF16 && (M == 256 || M == 512) && N >= 128 && K > 0 && K % 64 == 0;
}

extern "C" int smx_gemm_ln_pair_ok(int dtype, int N, int M, int K) {
  return smx_gemm_ln_fused_ok(dtype, N, M, K);
}

extern "C" size_t smx_gemm_colsum_workspace(int N, int M) {
  if (N <= 0 || M <= 0) return 0;
  return (size_t)((N + 63) / 64) * (size_t)M * sizeof(float);   // one partial row per N tile (64 = the small tile)
}

extern "C" int smx_gemm(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B,
                        int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K,
                        int batch, int splits, const smx_epilogue* epi, void* stream) {
  return gemm_impl(layout, dtype, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, N, M, K, batch, splits, 0, epi,
                   stream);
}

extern "C" int smx_gemm_plan_query(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb,
                                   int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K, int batch, int splits,
                                   const smx_epilogue* epi, smx_gemm_plan* plan) {
  SMX_REQUIRE(plan, "smx_gemm_plan_query: null plan");
  memset(plan, 0, sizeof(*plan));
  plan->kernel = -1;                                     // (nothing to launch: N == 0 or M == 0)
  return gemm_impl(layout, dtype, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, N, M, K, batch, splits, 0, epi, nullptr, nullptr,
                   nullptr, plan);
}

// ---- weight gradient: dW[b] (M x K) += alpha * dZ[b]^T X[b], reduce over `rows` frames ------------------------
// split-K over the frame dimension into fp32 slabs (plain coalesced stores), then ONE fixed-order reduction
// kernel adds the slabs into the gradient buffer: bit-reproducible, no atomics.
namespace smx {
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, long slab_stride,
                                                           float* dst, long lddst, long sdst, int M, int K, int batch,
                                                           float alpha, const float* __restrict__ bpart, float* dbias) {
  // bias gradient: dbias[b*M + m] += alpha * sum_s bpart[s][b*M + m]  (column sums of dZ collected by the wgrad GEMM).
  // 8 lanes share one element (slabs s = part, part+8, ..; 4 loads in flight each) and are folded in a fixed order:
  // a single thread walking up to 128 slabs is a 128-deep chain of dependent L2 round trips.
  if (dbias) {
    const long nb = (long)batch * M;
    const long nb8 = (nb * 8 + 255) / 256 * 256;          // whole wave iterations (shuffles need all 8 lanes present)
    for (long g = blockIdx.x * 256L + threadIdx.x; g < nb8; g += (long)gridDim.x * 256) {
      const long i = g >> 3;
      const int part = (int)(g & 7);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (i < nb) {
        int sidx = part;
        for (; sidx + 24 < nslab; sidx += 32) {
          a0 += bpart[(long)sidx * nb + i]; a1 += bpart[(long)(sidx + 8) * nb + i];
          a2 += bpart[(long)(sidx + 16) * nb + i]; a3 += bpart[(long)(sidx + 24) * nb + i];
        }
        for (; sidx < nslab; sidx += 8) a0 += bpart[(long)sidx * nb + i];
      }
      float a = (a0 + a1) + (a2 + a3);
      a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
      if (part == 0 && i < nb) dbias[i] += alpha * a;
    }
  }
  // weights: P lanes share one float4 of dW (slabs s = part, part+P, ..; 4 loads in flight each) and are folded by
  // shuffles in a fixed order - one thread walking all the slabs of a small weight (128 of them for a 256x256 matrix)
  // is a long chain of dependent L2 round trips.  P = 8 when there are many slabs, 1 for a handful.
  const int kv = K / 4;
  const long total = (long)batch * M * kv;
  const int P = nslab >= 16 ? 8 : 1, pshift = nslab >= 16 ? 3 : 0;
  const long totalp = ((total << pshift) + 255) / 256 * 256;   // whole wave iterations (shuffles need every lane)
  for (long g = blockIdx.x * 256L + threadIdx.x; g < totalp; g += (long)gridDim.x * 256) {
    const long i = g >> pshift;
    const int part = (int)(g & (P - 1));
    const bool ok = i < total;
    const long ic = ok ? i : 0;
    const int k4 = (int)(ic % kv);
    const long r = ic / kv;
    const int m = (int)(r % M), b = (int)(r / M);
    const float* sp = slabs + ((long)b * M + m) * K + k4 * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;   // 4 loads in flight, fixed order
    if (ok) {
      int sidx = part;
      for (; sidx + 3 * P < nslab; sidx += 4 * P) {
        const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)sidx * slab_stride);
        const float4 v1 = *reinterpret_cast<const float4*>(sp + (long)(sidx + P) * slab_stride);
        const float4 v2 = *reinterpret_cast<const float4*>(sp + (long)(sidx + 2 * P) * slab_stride);
        const float4 v3 = *reinterpret_cast<const float4*>(sp + (long)(sidx + 3 * P) * slab_stride);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; sidx < nslab; sidx += P) {
        const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)sidx * slab_stride);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
    }
    float sx = (a0.x + a1.x) + (a2.x + a3.x), sy = (a0.y + a1.y) + (a2.y + a3.y);
    float sz = (a0.z + a1.z) + (a2.z + a3.z), sw = (a0.w + a1.w) + (a2.w + a3.w);
    if (P == 8) {
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64);
        sz += __shfl_xor(sz, off, 64); sw += __shfl_xor(sw, off, 64);
      }
    }
    if (ok && part == 0) {
      float* d = dst + (long)b * sdst + (long)m * lddst + k4 * 4;
      d[0] += alpha * sx; d[1] += alpha * sy; d[2] += alpha * sz; d[3] += alpha * sw;
    }
  }
}
static int wgrad_splits(int rows, int M, int K, int batch) {
  long tiles = (long)((M + 127) / 128) * ((K + 127) / 128) * batch;
  // workgroups to aim for: exactly two per CU.  640 (2.5 per CU) leaves half the CUs with a third workgroup and the
  // launch takes as long as those; measured at 64000 frames: 71 -> 66 us (1024x256), 42 -> 35 us (256x256).
  // (The wide 128x256 tile does not help here: 2.5x slower with two register stages (spills), 71 vs 66 us with one.)
  // (with the LDS-DMA kernel and the wgrads on a side stream next to the dgrad chain, 384 = 1.5 per CU is the better
  // target for the step: C2b 27.04 -> 26.56 ms; fewer, longer splits also mean less slab traffic for the reduction)
  const long target = 384;
  long s = (target + tiles - 1) / tiles;
  // a whole split lives on one XCD (XCD x owns splits x, x + 8, ..): a split count that is not a multiple of 8 leaves
  // XCDs idle (6 splits of a 3072 x 512 weight: 236 us; 8 splits: 199 us)
  if (s >= 4) s = (s + 7) / 8 * 8;
  const int min_rows = 512;   // frames per split: fewer, longer splits when the batch is small (slab traffic)
  long smax = (rows + min_rows - 1) / min_rows;
  if (s > smax) s = smax;
  return (int)(s < 1 ? 1 : s);
}
}  // namespace smx

static int effective_splits(int K, int splits, int BK) {
  if (splits <= 1) return 1;
  int kc = (K + splits - 1) / splits;
  kc = ((kc + BK - 1) / BK) * BK;
  int s = (K + kc - 1) / kc;
  return s < 1 ? 1 : s;
}

extern "C" size_t smx_linear_wgrad_workspace(int rows, int M, int K, int batch) {
  return (size_t)wgrad_splits(rows, M, K, batch) * batch * ((size_t)M * K + M) * sizeof(float) + 16;
}

// slabs (+ bias partials) only; shared by the immediate and the deferred entry point
static int wgrad_slabs(int dtype, const void* dZ, int64_t lddz, int64_t strideZ, const void* X, int64_t ldx, int64_t strideX,
                       int rows, int M, int K, int batch, bool want_bias, float* ws, int* splits_out, void* stream,
                       const int* conv = nullptr) {
  smx_epilogue e;
  memset(&e, 0, sizeof(e));
  const int BK = dtype == SMX_BF16 ? 64 : 32;
  const int splits = effective_splits(rows, wgrad_splits(rows, M, K, batch), BK);
  e.alpha = 1.f;
  e.out_mode = SMX_OUT_F32;
  const long slab = (long)batch * M * K;
  float* bpart = want_bias ? ws + (long)splits * slab : nullptr;   // [splits][batch][M] behind the slabs
  *splits_out = splits;
  return gemm_impl(SMX_GEMM_TN, dtype, dZ, lddz, strideZ, X, ldx, strideX, ws, K, (int64_t)M * K, M, K, rows, batch, splits,
                   slab, &e, stream, bpart, conv);
}

extern "C" int smx_linear_wgrad(int dtype, const void* dZ, int64_t lddz, int64_t strideZ, const void* X, int64_t ldx,
                                int64_t strideX, float* dW, int64_t lddw, int64_t strideW, float* dbias, int rows, int M,
                                int K, int batch, float alpha, void* workspace, void* stream) {
  SMX_REQUIRE(dZ && X && dW, "smx_linear_wgrad: null pointer");
  SMX_REQUIRE(!dbias || (K % 4 == 0 && workspace && aligned16(workspace)),
              "smx_linear_wgrad: dbias needs the slab path (K %% 4 == 0 and an aligned workspace)");
  if (M <= 0 || K <= 0 || rows <= 0) return SMX_OK;
  if (K % 4 != 0 || workspace == nullptr || !aligned16(workspace)) {
    // ragged shapes: fp32 atomics straight into the gradient (not bit-reproducible)
    smx_epilogue e;
    memset(&e, 0, sizeof(e));
    const int BK = dtype == SMX_BF16 ? 64 : 32;
    const int splits = effective_splits(rows, wgrad_splits(rows, M, K, batch), BK);
    e.alpha = alpha;
    e.out_mode = SMX_OUT_ATOMIC_F32;
    return gemm_impl(SMX_GEMM_TN, dtype, dZ, lddz, strideZ, X, ldx, strideX, dW, lddw, strideW, M, K, rows, batch, splits,
                     0, &e, stream);
  }
  float* ws = reinterpret_cast<float*>(workspace);
  int splits = 1;
  int rc = wgrad_slabs(dtype, dZ, lddz, strideZ, X, ldx, strideX, rows, M, K, batch, dbias != nullptr, ws, &splits, stream);
  if (rc != SMX_OK) return rc;
  const long slab = (long)batch * M * K;
  float* bpart = dbias ? ws + (long)splits * slab : nullptr;
  long total = (long)batch * M * (K / 4) * (splits >= 16 ? 8 : 1);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ws,
                     splits, slab, dW, lddw, strideW, M, K, batch, alpha, bpart, dbias);
  return check_launch("smx_linear_wgrad");
}

// ---- the front-end's second conv block (64 -> O channels, 3 x 3, stride 2, reflect pad 1) WITHOUT the patch matrix: the
// GEMM kernels gather their operand from the channels-last input (gemm_kernel<..., GATHER>, one tap per K tile / column tile)
// DFT of overlapping frames of a zero-padded waveform, window folded into the bases (symmetric window): see fold_stage_load
extern "C" int smx_dft_frames(const float* wav_padded, int64_t ldw, const float* basis_cos, const float* basis_sin, float* spec,
                              int64_t lds, int im_off, int B, int T, int n_fft, int hop, int rows_basis, void* stream) {
  SMX_REQUIRE(wav_padded && basis_cos && basis_sin && spec && n_fft % 8 == 0 && hop % 4 == 0 && rows_basis % 4 == 0 &&
                  rows_basis <= im_off && lds >= im_off + rows_basis && lds % 4 == 0 && ldw % 4 == 0 && ldw >= (int64_t)(T - 1) * hop + n_fft + 4,
              "smx_dft_frames: bad arguments");
  if (B <= 0 || T <= 0) return SMX_OK;
  smx_epilogue e;
  memset(&e, 0, sizeof(e));
  e.alpha = 1.f;
  const int kc = n_fft / 2 + 4, ks = n_fft / 2;
  const int cv[2] = {-3, n_fft}, sv[2] = {-4, n_fft};
  int rc = gemm_impl(SMX_GEMM_NT, SMX_F32, wav_padded, hop, ldw, basis_cos, kc, 0, spec, lds, (int64_t)T * lds, T, rows_basis, kc, B, 1, 0,
                     &e, stream, nullptr, cv);
  if (rc != SMX_OK) return rc;
  return gemm_impl(SMX_GEMM_NT, SMX_F32, wav_padded, hop, ldw, basis_sin, ks, 0, spec + im_off, lds, (int64_t)T * lds, T, rows_basis, ks, B, 1,
                   0, &e, stream, nullptr, sv);
}

extern "C" int smx_conv2d_s2_fwd(int dtype, const void* X, const void* Wg, const float* bias, void* Y, int B, int T, int F, int C,
                                 int O, int Kp, void* stream) {
  SMX_REQUIRE(X && Wg && Y && T >= 2 && F >= 2 && Kp >= 9 * C, "smx_conv2d_s2_fwd: bad arguments");
  if (dtype != SMX_BF16 || C != 64 || O % 8 != 0) return fail(SMX_EUNSUPPORTED, "smx_conv2d_s2_fwd: built for bf16, C = 64, O %% 8 == 0");
  if (B <= 0) return SMX_OK;
  smx_epilogue e;
  memset(&e, 0, sizeof(e));
  e.alpha = 1.f;
  e.bias = bias;
  const int conv[2] = {T, F};
  const long rows = (long)B * ((T + 1) / 2) * ((F + 1) / 2);
  SMX_REQUIRE(rows < (1L << 31), "smx_conv2d_s2_fwd: too many output pixels");
  return gemm_impl(SMX_GEMM_NT, dtype, X, 9 * C, 0, Wg, Kp, 0, Y, O, 0, (int)rows, O, 9 * C, 1, 1, 0, &e, stream, nullptr, conv);
}
extern "C" size_t smx_conv2d_s2_wgrad_workspace(int B, int T, int F, int C, int O) {
  const long rows = (long)B * ((T + 1) / 2) * ((F + 1) / 2);
  return smx_linear_wgrad_workspace((int)rows, O, 9 * C, 1);
}
extern "C" int smx_conv2d_s2_wgrad(int dtype, const void* dY, const void* X, float* dWg, float* dbias, int B, int T, int F, int C,
                                   int O, int Kp, void* workspace, void* stream) {
  SMX_REQUIRE(dY && X && dWg && workspace && aligned16(workspace) && T >= 2 && F >= 2 && Kp >= 9 * C, "smx_conv2d_s2_wgrad: bad arguments");
  if (dtype != SMX_BF16 || C != 64 || O % 8 != 0) return fail(SMX_EUNSUPPORTED, "smx_conv2d_s2_wgrad: built for bf16, C = 64, O %% 8 == 0");
  if (B <= 0) return SMX_OK;
  const long rows = (long)B * ((T + 1) / 2) * ((F + 1) / 2);
  const int conv[2] = {T, F}, K = 9 * C;
  float* ws = reinterpret_cast<float*>(workspace);
  int splits = 1;
  int rc = wgrad_slabs(dtype, dY, O, 0, X, K, 0, (int)rows, O, K, 1, dbias != nullptr, ws, &splits, stream, conv);
  if (rc != SMX_OK) return rc;
  const long slab = (long)O * K;
  float* bpart = dbias ? ws + (long)splits * slab : nullptr;
  long total = (long)O * (K / 4) * (splits >= 16 ? 8 : 1);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ws, splits,
                     slab, dWg, (long)Kp, (long)O * Kp, O, K, 1, 1.f, bpart, dbias);
  return check_launch("smx_conv2d_s2_wgrad");
}

extern "C" int smx_linear_wgrad_partial(int dtype, const void* dZ, int64_t lddz, int64_t strideZ, const void* X, int64_t ldx,
                                        int64_t strideX, int rows, int M, int K, int batch, int want_bias, void* workspace,
                                        int32_t* nslabs, int64_t* slab_stride, int64_t* bias_offset, void* stream) {
  SMX_REQUIRE(dZ && X && workspace && nslabs && slab_stride && bias_offset, "smx_linear_wgrad_partial: null pointer");
  if (K % 4 != 0 || !aligned16(workspace) || M <= 0 || K <= 0 || rows <= 0)
    return fail(SMX_EUNSUPPORTED, "smx_linear_wgrad_partial: needs K %% 4 == 0, positive sizes and an aligned workspace");
  int splits = 1;
  int rc = wgrad_slabs(dtype, dZ, lddz, strideZ, X, ldx, strideX, rows, M, K, batch, want_bias != 0,
                       reinterpret_cast<float*>(workspace), &splits, stream);
  if (rc != SMX_OK) return rc;
  *nslabs = splits;
  *slab_stride = (int64_t)batch * M * K;
  *bias_offset = (int64_t)splits * batch * M * K;
  return SMX_OK;
}

extern "C" int smx_linear_act_mask_fwd(int dtype, const void* X, int64_t ldx, const void* W, int64_t ldw, void* Y,
                                       int64_t ldy, int N, int M, int K, const smx_epilogue* epi, void* stream) {
  return smx_gemm(SMX_GEMM_NT, dtype, X, ldx, 0, W, ldw, 0, Y, ldy, 0, N, M, K, 1, 1, epi, stream);
}
