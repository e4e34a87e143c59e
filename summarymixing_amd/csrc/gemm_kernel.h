// gemm_kernel.h — the tiled MFMA GEMM kernel template of libsmx.so (gfx950): every tile / layout / epilogue instantiation of
// gemm.hip (the plain tiles), gemm_ln256.hip and gemm_ln512.hip (the LayerNorm-fused row-complete tiles; separate translation
// units so that the three compile in parallel).  Design notes: the header comment of gemm.hip.
#pragma once
#include <utility>

#include "gemm_common.h"

namespace smx {

// ---- the kernel ---------------------------------------------------------------------------------------------
// tuning constants of the register-staged tiles (swept in rounds 1-4)
#ifndef SMX_KOCC
#define SMX_KOCC 3
#endif
#ifndef SMX_R64_OCC
#define SMX_R64_OCC 2      // workgroups per CU of the 64 x 256 LayerNorm-fused tile (3: 168 registers)
#endif
#ifndef SMX_NS_KC
#define SMX_NS_KC 1
#endif
constexpr int kOcc = SMX_KOCC;        // workgroups per CU of the 128 x 128 / 64 x 64 tiles (168-register budget)
constexpr int kNsSmall = 4;    // register stages in flight of the 64 x 64 tile (latency-bound small grids: see the main loop)
constexpr int kNsKc = SMX_NS_KC;       // ... of the 128 x 128 tile

// wgrad (TN) runs exactly two workgroups per CU (wgrad_splits), so it takes the 256-register budget: no spills with
// the bias-gradient column sums and two register stages of both operands; the LDS-DMA variant's ring is 64 KB.
// LNF: 0 = ordinary epilogue; 1 = LayerNorm backward fused (SMX_EPI_LN_BWD; 3 = with its activation extensions),
// 2 = LayerNorm forward appended (SMX_EPI_LN_FWD) - separate instantiations of the 128 x 256 bf16 kernel, so that their extra live registers never cost
// the ordinary one anything.
// GATHER (bf16, 64 x 64 tile): 1 = the A operand of an NT GEMM, 2 = the B operand of a TN GEMM is the implicit patch matrix of a
// 3 x 3 / stride 2 convolution over 64 channels (GemmParams::g_*): the front-end's second block without im2col.
template <typename T, bool A_KC, bool B_KC, int TILE_N, int TILE_M, bool VEC, int LNF = 0, int GATHER = 0>
__global__ __launch_bounds__(256, ((TILE_N == 256 || TILE_M == 512) ? 1 : ((TILE_N == 64 && TILE_M == 256) ? SMX_R64_OCC : ((TILE_M > 128 || !A_KC) ? 2 : kOcc)))) void gemm_kernel(GemmParams p) {
  static_assert(GATHER == 0 || GATHER >= 3 || (sizeof(T) == 2 && VEC && TILE_N == 64 && TILE_M == 64 && LNF == 0), "GATHER 1 / 2: bf16 64 x 64 tile");
  static_assert(GATHER < 3 || (sizeof(T) == 4 && A_KC && B_KC && LNF == 0), "GATHER 3 / 4 (folded DFT frames): float32 NT");
  static_assert(GATHER != 1 || (A_KC && B_KC), "GATHER 1: NT");
  static_assert(GATHER != 2 || (!A_KC && !B_KC), "GATHER 2: TN");
  static_assert(LNF == 0 || (sizeof(T) == 2 && VEC && (TILE_M == 256 || TILE_M == 512) && (TILE_N == 128 || (TILE_N == 64 && TILE_M == 256))),
                "fused LayerNorm: bf16 128 x 256 / 128 x 512 / 64 x 256 tile");
  static_assert(LNF <= 3 || LNF == 5 || LNF == 7, "LNF: 1 / 3 LayerNorm backward (3: extended), 2 forward, +4 = float32 ln_x");
  // ROW512: the row-complete tile of d_model = 512 (128 rows x 512 columns = 64 K outputs, ONE workgroup per CU, the four waves side
  // by side: 128 x 128 outputs = 256 accumulator registers each, the wave tile of the 256 x 256 kernel) on the same explicit
  // software pipeline (T256P below); its two 80 KB operand stages fill the CU's 160 KB of LDS, the epilogue arrays alias them
  constexpr bool ROW512 = TILE_M == 512;
  static_assert(!ROW512 || (sizeof(T) == 2 && VEC && A_KC && TILE_N == 128 && GATHER == 0), "128 x 512 tile: bf16, aligned, NT / NN");
  constexpr int BK = ElemTraits<T>::BK;
  // ROW64 (round 6): the row-complete 64 x 256 tile of d_model = 256 for 17 500 - 37 000 frames, where 128-row tiles are fewer than the
  // chip's 512 workgroup slots (B = 64 x 500: 250): twice the workgroups, the four waves side by side (64 x 64 outputs each: as many
  // fragment reads per MFMA as the 128 x 256 tile's 64 x 128), the same software pipeline with ONE activation piece per step
  constexpr bool ROW64 = TILE_N == 64 && TILE_M == 256;
  static_assert(!ROW64 || (sizeof(T) == 2 && VEC && A_KC && GATHER == 0), "64 x 256 tile: bf16, aligned, NT / NN");
  constexpr int WAVES_M = (ROW512 || ROW64) ? 4 : 2, WAVES_N = 4 / WAVES_M;
  constexpr int WN = TILE_N / WAVES_N, WM = TILE_M / WAVES_M;
  constexpr int FN = WN / 32, FM = WM / 32;
  constexpr int A_BYTES = lds_bytes<T, TILE_N, A_KC>();
  // (ROW512, reduce-strided weights: the swizzled image without a row pad, frag_tr_swz_rb - 64 k rows of 1 KB)
  constexpr int B_BYTES = (ROW512 && !B_KC) ? 64 * TILE_M * 2 : lds_bytes<T, TILE_M, B_KC>();
  // the epilogue stages PH_ROWS fp32 rows at a time: half a tile, or 32 rows for the wide (TILE_M = 256) tile so that
  // the block stays under 64 KB of LDS
  constexpr int PH_FRAGS = TILE_M > 128 ? 1 : FN;                // 32-row accumulator fragments per phase
  constexpr int PH_ROWS = 32 * PH_FRAGS;
  constexpr int NPH = TILE_N / PH_ROWS;
  constexpr int EPI_BYTES = PH_ROWS * (TILE_M * 4 + 16);         // fp32 rows, 16 B row pad
  // T256P (256 x 256 bf16 tile, one workgroup of four waves per CU): software-pipelined main loop with double-buffered LDS stages
  constexpr bool T256P = (sizeof(T) == 2 && VEC && A_KC && TILE_N == 256 && TILE_M == 256 && GATHER == 0 && LNF == 0) || ROW512;
  // W128P: the 128 x 256 tile (two workgroups per CU, every LayerNorm-fused epilogue) with the same explicit software pipeline
  // in 32-element steps: both operands double-buffered in LDS (2 x 24 KB = the one 48 KB stage of the serial loop)
  constexpr bool W128P = sizeof(T) == 2 && VEC && A_KC && (TILE_N == 128 || TILE_N == 64) && TILE_M == 256 && GATHER == 0;
  constexpr int AB_BYTES = T256P ? 2 * (A_BYTES + B_BYTES) : A_BYTES + B_BYTES;
  constexpr int SMEM_BYTES = AB_BYTES > EPI_BYTES ? AB_BYTES : EPI_BYTES;
  constexpr int RED_BYTES = TILE_M * 4;                          // colsum: phase-0 column sums
  // bias[TILE_M] | row factors[TILE_N] (mask * alpha) [| LN gamma | beta [LN forward: | the tile's (mean, rstd) pairs of both
  // LayerNorms | gamma | beta of the second LayerNorm]]
  constexpr int SIDE_BYTES = (TILE_M + TILE_N + (LNF ? 2 * TILE_M : 0) + (LNF == 2 ? 4 * 128 + 2 * TILE_M : 0)) * 4;   // (statistics block: [2][128][2] whatever TILE_N)
  // when the block would pass the 64 KB static LDS limit its small epilogue arrays live behind the epilogue staging rows
  // inside the (by then dead) operand stage, fenced by one extra barrier
  constexpr bool ALIAS_SIDE = SMEM_BYTES + RED_BYTES + SIDE_BYTES > 65536;
  static_assert(!ALIAS_SIDE || EPI_BYTES + RED_BYTES + SIDE_BYTES + 64 <= SMEM_BYTES, "epilogue arrays do not fit");
  __shared__ __attribute__((aligned(16))) char smem[ALIAS_SIDE ? SMEM_BYTES : SMEM_BYTES + RED_BYTES + SIDE_BYTES];
  float* red = reinterpret_cast<float*>(smem + (ALIAS_SIDE ? (EPI_BYTES + 63) / 64 * 64 : SMEM_BYTES));
  float* side = red + TILE_M;
  char* As = smem;
  char* Bs = smem + A_BYTES;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave / WAVES_M, wm = wave % WAVES_M;
#ifdef SMX_DIAG   // per-wave clock stamps (tools/gemm_stamps.py with libsmx_diag.so); the product kernel carries none
  long long* dbgp = p.dbg ? p.dbg + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 : nullptr;
#define SMX_STAMP(k) do { if (dbgp && lane == 0) dbgp[k] = clock64(); } while (0)
#else
#define SMX_STAMP(k) do { } while (0)
#endif
  SMX_STAMP(0);
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware work mapping (workgroup b runs on XCD b % 8, each XCD has its own L2):
  //  * splits == 1: consecutive remapped ids walk the M tiles of one A row panel, so a panel is fetched into ONE L2;
  //  * split-K (wgrad): all tiles of one K-range read the same operand rows, so a whole split is given to one XCD
  //    (XCD x owns splits x, x+8, ...).  With the naive (tile, split) grid every XCD streamed every input row:
  //    8x the HBM/fabric traffic.
  int tile_n, tile_m, bz, split;
  {
    const int ntiles = p.tiles_n * p.tiles_m;
    if (p.splits == 1) {
      int bid = blockIdx.x;
      const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      tile_n = bid / p.tiles_m; tile_m = bid % p.tiles_m;
      bz = blockIdx.y; split = 0;
    } else {
      const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
      const int per = ntiles * p.batch;                 // work items of one split
      const int item = idx % per;
      split = (idx / per) * 8 + xcd;
      if (split >= p.splits) return;
      bz = item / ntiles;
      const int tl = item % ntiles;
      tile_n = tl / p.tiles_m; tile_m = tl % p.tiles_m;
    }
  }
  const int n0 = tile_n * TILE_N, m0 = tile_m * TILE_M;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const T* A = reinterpret_cast<const T*>(p.A) + (long)bz * p.sA;
  const T* B = reinterpret_cast<const T*>(p.B) + (long)bz * p.sB;
  // operand stage loads: buffer loads with once-computed offsets for the aligned bf16 kernels (BufStage), else the
  // generic guarded loads
  constexpr bool BUFLD = sizeof(T) == 2 && VEC;
  BufStage<T, A_KC, TILE_N> bufa;
  BufStage<T, B_KC, TILE_M> bufb;
  GatherStageKC<GATHER == 1 ? TILE_N : 32> gka;
  GatherStageKS<GATHER == 2 ? TILE_M : 32> gkb;
  if constexpr (BUFLD) {
    if constexpr (GATHER == 1) gka.init(p.A, p.g_npix, p, n0, t);
    else bufa.init(A, p.lda, n0, p.N, p.K, t);
    if constexpr (GATHER == 2) gkb.init(p.B, p.g_npix, m0, t);
    else bufb.init(B, p.ldb, m0, p.M, p.K, t);
  }
  auto load_a = [&](uint4 (&reg)[TILE_N / 32], int k0) {
    if constexpr (GATHER == 1) gka.load(reg, k0, p);
    else if constexpr (GATHER >= 3) fold_stage_load<GATHER, TILE_N>(reg, reinterpret_cast<const float*>(A), p.lda, n0, p.N, k0, p.g_T, t);
    else if constexpr (BUFLD) bufa.load(reg, k0);
    else stage_load<T, A_KC, TILE_N, VEC>(reg, A, p.lda, n0, p.N, k0, kend, t);
  };
  auto load_b = [&](uint4 (&reg)[TILE_M / 32], int k0) {
    if constexpr (GATHER == 2) gkb.load(reg, k0, kend, p);
    else if constexpr (BUFLD) bufb.load(reg, k0);
    else stage_load<T, B_KC, TILE_M, VEC>(reg, B, p.ldb, m0, p.M, k0, kend, t);
  };

  // epilogue side vector of this thread (requested first, parked in ONE register across the main loop, published to
  // LDS before the epilogue): t < TILE_M -> bias[m0 + t], then TILE_N row factors row_mask[n] * alpha
  constexpr int NSIDE = (TILE_M + TILE_N + 255) / 256;
  // Narrow tiles (SIDE_RAW): the RAW loaded words stay parked - bias bits and mask byte in SEPARATE registers - and are
  // converted where they are published: converting here, or letting the two divergent branches write one register, made every
  // workgroup wait for that round trip, s_waitcnt vmcnt(0), before it requested its first operand tile (what a small grid - the
  // recipe batch's 472 workgroups - cannot hide).  The 256-wide tiles have no register to spare for it (the 256 x 256 tile: 16 ->
  // 147 spilled registers) and keep the converted value in ONE register.
  constexpr bool SIDE_RAW = TILE_M <= 128;
  float side_b[NSIDE];
  uint32_t side_m[SIDE_RAW ? NSIDE : 1];
#pragma unroll
  for (int i = 0; i < NSIDE; ++i) {
    const int si = t + 256 * i, n = n0 + si - TILE_M;
    const bool on = p.e.out_mode != SMX_OUT_ATOMIC_F32;
    if constexpr (SIDE_RAW) {
      side_b[i] = (on && si < TILE_M && p.e.bias && m0 + si < p.M) ? p.e.bias[(long)bz * p.e.bias_batch_stride + m0 + si] : 0.f;
      side_m[i] = (on && si >= TILE_M && si < TILE_M + TILE_N && p.e.row_mask && n < p.N) ? (uint32_t)p.e.row_mask[n] : 1u;
    } else {
      side_b[i] = 0.f;
      if (on) {
        if (si < TILE_M) {
          if (p.e.bias && m0 + si < p.M) side_b[i] = p.e.bias[(long)bz * p.e.bias_batch_stride + m0 + si];
        } else if (si < TILE_M + TILE_N) {
          side_b[i] = ((p.e.row_mask && n < p.N) ? (p.e.row_mask[n] ? 1.f : 0.f) : 1.f) * p.e.alpha;
        }
      }
    }
  }
  auto side_value = [&](int i) __attribute__((always_inline)) -> float {
    if constexpr (SIDE_RAW) {
      if (p.e.out_mode == SMX_OUT_ATOMIC_F32) return 0.f;
      asm volatile("" : "+v"(side_m[i]));                // (keeps hipcc from moving the compare up behind the load)
      return t + 256 * i < TILE_M ? side_b[i] : (side_m[i] ? p.e.alpha : 0.f);
    } else {
      return side_b[i];
    }
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint32_t fpa[FN], fpb[FM];                             // loop-invariant fragment addresses (reduce-contiguous images)
#pragma unroll
  for (int i = 0; i < FN; ++i) fpa[i] = frag_pre(wn * WN + i * 32 + l31, hi);
#pragma unroll
  for (int j = 0; j < FM; ++j) fpb[j] = frag_pre(wm * WM + j * 32 + l31, hi);
#ifdef SMX_DIAG
  const bool ab_nold = p.ablate & 4, ab_nomfma = p.ablate & 2, ab_nost = p.ablate & 1;   // tools/one_gemm.py phase ablations
#else
  constexpr bool ab_nold = false, ab_nomfma = false, ab_nost = false;                     // (compiled out of the product)
#endif
  constexpr int CSN = 16 / (int)sizeof(T);
  float cs[CSN];
#pragma unroll
  for (int q = 0; q < CSN; ++q) cs[q] = 0.f;
  const bool do_cs = !A_KC && p.acolsum != nullptr && tile_m == 0;
  if constexpr (W128P) {
    // ---- 128 x 256 tile, explicit software pipeline (the structure that worked on the 256 x 256 tile, T256P below), 32 reduce
    // elements per step: step h multiplies half-stage h out of LDS buffer h & 1 and, between its 16 MFMAs, moves half-stage
    // h + 1 from registers into the other buffer (2 + 4 ds_write_b128), refills those registers (activations two steps ahead,
    // weights one) and reads the second sub-step's fragments.  ONE barrier per step; sched_barrier(0) after every MFMA pins the
    // order.  LDS images of a half-stage: reduce-contiguous operands [rows][32 k] = 64-byte rows, 16-byte chunk c of row r at
    // position c ^ ((r >> 2) & 3) (conflict-free ds_read_b128); the reduce-strided weights of NN as they lie, [32 k][256 + 32].
    typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
    constexpr int HK = 32, AH = TILE_N * HK * 2, BH = B_KC ? TILE_M * HK * 2 : HK * (TILE_M + 32) * 2, NPA = TILE_N / 64, NPBW = TILE_M / 64;
    constexpr int RC = TILE_M / 8;                        // 16-byte chunks per k row of a reduce-strided weight half-stage
    static_assert(2 * (AH + BH) <= AB_BYTES, "two half-stage pairs fit the serial loop's stage");
    char* Abuf = smem;
    char* Bbuf = smem + 2 * AH;
    const int nh = (kend - kbeg) / HK;
    // this thread's pieces: v = t + 256 i
    uint32_t va[NPA], vbw[NPBW];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int rg = n0 + ((t + 256 * i) >> 2);
      va[i] = rg < p.N ? (uint32_t)(((long)rg * p.lda + (t & 3) * 8) * 2) : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < NPBW; ++i) {
      if constexpr (B_KC) {
        const int rg = m0 + ((t + 256 * i) >> 2);
        vbw[i] = rg < p.M ? (uint32_t)(((long)rg * p.ldb + (t & 3) * 8) * 2) : 0x80000000u;
      } else {
        const int cg = m0 + (t % RC) * 8;
        vbw[i] = cg < p.M ? (uint32_t)(((long)(t / RC + (256 / RC) * i) * p.ldb + cg) * 2) : 0x80000000u;
      }
    }
    const uint32_t a_st = (uint32_t)((t >> 2) * 64 + (((t & 3) ^ ((t >> 4) & 3)) << 4));
    const uint32_t b_st = B_KC ? a_st : (uint32_t)((t / RC) * ((TILE_M + 32) * 2) + (t % RC) * 16);
    constexpr uint32_t A_PIECE = 64 * 64, B_PIECE = B_KC ? 64 * 64 : (256 / RC) * (TILE_M + 32) * 2;
    const uint32_t kba = HK * 2, kbb = B_KC ? HK * 2 : (uint32_t)(HK * p.ldb * 2);   // bytes per half-stage along k
    auto ld_a = [&](int h, auto itag) __attribute__((always_inline)) {
      constexpr int I = decltype(itag)::value;
      const bool v_ = h < nh;
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(bufa.rsrc, va[I] | (v_ ? 0u : 0x80000000u), v_ ? (uint32_t)(kbeg / HK + h) * kba : 0u, 0);
      return make_uint4(r.x, r.y, r.z, r.w);
    };
    auto ld_b = [&](int h, auto itag) __attribute__((always_inline)) {
      constexpr int I = decltype(itag)::value;
      const bool v_ = h < nh;
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(bufb.rsrc, vbw[I] | (v_ ? 0u : 0x80000000u), v_ ? (uint32_t)(kbeg / HK + h) * kbb : 0u, 0);
      return make_uint4(r.x, r.y, r.z, r.w);
    };
    uint32_t fqa[FN], fqb[FM];                           // loop-invariant fragment addresses (64-byte rows)
#pragma unroll
    for (int i = 0; i < FN; ++i) { const int r = wn * WN + i * 32 + l31; fqa[i] = (uint32_t)(r * 64 + ((hi ^ ((r >> 2) & 3)) << 4)); }
#pragma unroll
    for (int j = 0; j < FM; ++j) { const int r = wm * WM + j * 32 + l31; fqb[j] = (uint32_t)(r * 64 + ((hi ^ ((r >> 2) & 3)) << 4)); }
    uint4 ra[2][NPA], rb[NPBW];
    for_seq<0, NPA>([&](auto i) __attribute__((always_inline)) { ra[0][decltype(i)::value] = ld_a(0, i); });
    for_seq<0, NPA>([&](auto i) __attribute__((always_inline)) { ra[1][decltype(i)::value] = ld_a(1, i); });
    for_seq<0, NPBW>([&](auto i) __attribute__((always_inline)) { rb[decltype(i)::value] = ld_b(0, i); });
    for_seq<0, NPA>([&](auto i) __attribute__((always_inline)) { *reinterpret_cast<uint4*>(Abuf + a_st + decltype(i)::value * A_PIECE) = ra[0][decltype(i)::value]; });
    for_seq<0, NPBW>([&](auto i) __attribute__((always_inline)) { *reinterpret_cast<uint4*>(Bbuf + b_st + decltype(i)::value * B_PIECE) = rb[decltype(i)::value]; });
    for_seq<0, NPA>([&](auto i) __attribute__((always_inline)) { ra[0][decltype(i)::value] = ld_a(2, i); });
    for_seq<0, NPBW>([&](auto i) __attribute__((always_inline)) { rb[decltype(i)::value] = ld_b(1, i); });
    lds_barrier();
    SMX_STAMP(1);
    auto step = [&](int h, auto utag) __attribute__((always_inline)) {
      constexpr int U = decltype(utag)::value;                                      // h & 1
      const char* Ab = Abuf + U * AH;
      const char* Bb = Bbuf + U * BH;
      char* An = Abuf + (U ^ 1) * AH;
      char* Bn = Bbuf + (U ^ 1) * BH;
      bf16x8 fa[2][FN], fb[2][FM];
      auto read_frag = [&](int kk, int buf, auto ftag) __attribute__((always_inline)) {
        constexpr int Fi = decltype(ftag)::value;
        if constexpr (Fi < FN) fa[buf][Fi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Ab + (fqa[Fi] ^ (uint32_t)(kk << 5))));
        else if constexpr (B_KC) fb[buf][Fi - FN] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bb + (fqb[Fi - FN] ^ (uint32_t)(kk << 5))));
        else fb[buf][Fi - FN] = frag_bf16<B_KC, TILE_M>(Bb, wm * WM + (Fi - FN) * 32 + l31, kk, hi);
      };
      auto move_piece = [&](auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        if constexpr (P < NPA) {
          *reinterpret_cast<uint4*>(An + a_st + P * A_PIECE) = ra[U ^ 1][P];
          ra[U ^ 1][P] = ld_a(h + 3, ActTag<P>{});
        } else {
          *reinterpret_cast<uint4*>(Bn + b_st + (P - NPA) * B_PIECE) = rb[P - NPA];
          rb[P - NPA] = ld_b(h + 2, ActTag<P - NPA>{});
        }
      };
      for_seq<0, FN + FM>([&](auto f) __attribute__((always_inline)) { read_frag(0, 0, f); });
      __builtin_amdgcn_sched_barrier(0);
      for_seq<0, FN * FM * 2>([&](auto stag) __attribute__((always_inline)) {
        constexpr int S = decltype(stag)::value, kk = S / (FN * FM), q = S % (FN * FM), i = q / FM, j = q % FM;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
        if constexpr (kk == 0 && q < FN + FM) read_frag(1, 1, ActTag<q>{});
        // piece p goes behind MFMA slot (p + 1) * NSL / (NP + 1): spread over the step, none behind the last MFMA
        constexpr int NSL = FN * FM * 2, NP = NPA + NPBW;
        for_seq<0, NP>([&](auto pt) __attribute__((always_inline)) {
          constexpr int P = decltype(pt)::value;
          if constexpr ((P + 1) * NSL / (NP + 1) == S) move_piece(pt);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      lds_barrier();
    };
    for (int h = 0; h + 1 < nh; h += 2) {
      step(h, ActTag<0>{});
      step(h + 1, ActTag<1>{});
    }
    if (nh & 1) step(nh - 1, ActTag<0>{});
  } else if constexpr (T256P) {
    // ---- 256 x 256 tile: ONE wave per SIMD (128 x 128 outputs = 256 accumulator registers), nobody else hides its latencies,
    // so the K loop is software-pipelined the way the vendor library's is.  Step h multiplies stage h out of LDS buffer h & 1
    // and, between its 64 MFMAs, (i) moves stage h + 1 from registers into the OTHER buffer, (ii) refills those registers from
    // memory (activations two stages ahead, weights - L2 resident - one), (iii) reads the fragments of the next 16-element
    // sub-step.  One barrier per step.  The instruction order IS the source order (for_seq + sched_barrier(0) after every MFMA):
    // per MFMA at most one fragment read, the stage's pieces (one ds_write_b128 + one buffer load each) spread evenly over the step.
    // Per 64 reduce elements and CU: 64 KB through the vector-memory pipe and into LDS, 128 KB of fragment reads, for
    // 2048 cycles of MFMA issue per SIMD (two 128 x 256 workgroups: 96 KB / 192 KB for the same flops).
    // ROW512 (128 x 512, waves side by side): the same wave tile and step; 80 KB staged per step (4 activation + 16 weight pieces
    // per thread), every activation fragment is read by all four waves.
    char* Abuf = smem;
    char* Bbuf = smem + 2 * A_BYTES;
    const int nk = (kend - kbeg) / BK;
    uint4 ra[2][TILE_N / 32], rb[TILE_M / 32];
    bufa.load_pred(ra[0], kbeg, nk > 0 );
    bufa.load_pred(ra[1], kbeg + BK, nk > 1 );
    bufb.load_pred(rb, kbeg, nk > 0 );
    // piece p of a stage = the p-th 16-byte vector of this thread (activations first, then weights): its LDS address is a base
    // plus p times a constant (stage_store: v = t + 256 p), so a piece is one ds_write_b128 / one buffer_load_dwordx4.
    // Reduce-strided weights: 256 columns as they lie with a 64-byte row pad (frag_bf16); 512 columns (ROW512) without a pad,
    // 16-byte granule g of k row k at g ^ (4 * (k & 3)) (frag_tr_swz_rb: thread t owns granule t & 63 of k rows (t >> 6) + 4 q)
    const uint32_t a_st = (uint32_t)((t >> 3) * 128 + (((t & 7) ^ ((t >> 4) & 7)) << 4));
    const uint32_t b_st = B_KC ? a_st
                               : (ROW512 ? (uint32_t)((t >> 6) * (TILE_M * 2) + (((t & 63) ^ (((t >> 6) & 3) << 2)) << 4))
                                         : (uint32_t)((t >> 5) * ((TILE_M + 32) * 2) + (t & 31) * 16));
    constexpr uint32_t A_PIECE = 32 * 128, B_PIECE = B_KC ? 32 * 128 : (ROW512 ? 4 * TILE_M * 2 : 8 * (TILE_M + 32) * 2);
    constexpr int NPA = TILE_N / 32, NPB = TILE_M / 32, NP = NPA + NPB, NSLOT = FN * FM * (BK / 16);
    static_assert(NSLOT == 64 && (NP == 16 || NP == 20), "64 MFMAs per step; 16 pieces (256 x 256) or 20 (128 x 512)");
    stage_store<T, true, TILE_N>(ra[0], Abuf, t);
    if constexpr (ROW512 && !B_KC) {
      for_seq<0, NPB>([&](auto qt) __attribute__((always_inline)) { *reinterpret_cast<uint4*>(Bbuf + b_st + decltype(qt)::value * B_PIECE) = rb[decltype(qt)::value]; });
    } else {
      stage_store<T, B_KC, TILE_M>(rb, Bbuf, t);
    }
    bufa.load_pred(ra[0], kbeg + 2 * BK, nk > 2 );
    bufb.load_pred(rb, kbeg + BK, nk > 1 );
    lds_barrier();
    SMX_STAMP(1);
    auto step = [&](int h, auto utag) __attribute__((always_inline)) {
      constexpr int U = decltype(utag)::value;                                      // h & 1
      const char* Ab = Abuf + U * A_BYTES;
      const char* Bb = Bbuf + U * B_BYTES;
      char* An = Abuf + (U ^ 1) * A_BYTES;
      char* Bn = Bbuf + (U ^ 1) * B_BYTES;
      const uint32_t soa = (h + 3 < nk) ? (uint32_t)(kbeg + (h + 3) * BK) * bufa.kbytes : 0u, ina = (h + 3 < nk) ? 0u : 0x80000000u;
      const uint32_t sob = (h + 2 < nk) ? (uint32_t)(kbeg + (h + 2) * BK) * bufb.kbytes : 0u, inb = (h + 2 < nk) ? 0u : 0x80000000u;
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
      // one piece: stage h + 1 from its register into the other LDS buffer, then the register's refill from memory
      auto move_piece = [&](auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        if constexpr (P < NPA) {
          *reinterpret_cast<uint4*>(An + a_st + P * A_PIECE) = ra[U ^ 1][P];
          const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(bufa.rsrc, bufa.voff[P] | ina, soa, 0);
          ra[U ^ 1][P] = make_uint4(r.x, r.y, r.z, r.w);
        } else {
          constexpr int Q = P - NPA;
          *reinterpret_cast<uint4*>(Bn + b_st + Q * B_PIECE) = rb[Q];
          const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(bufb.rsrc, bufb.voff[Q] | inb, sob, 0);
          rb[Q] = make_uint4(r.x, r.y, r.z, r.w);
        }
      };
      // fragment f of sub-step kk: 0..FN-1 activations, FN.. weights
      bf16x8 fa[2][FN], fb[2][FM];
      auto read_frag = [&](int kk, int buf, auto ftag) __attribute__((always_inline)) {
        constexpr int Fi = decltype(ftag)::value;
        if constexpr (Fi < FN) fa[buf][Fi] = frag_kc(Ab, fpa[Fi], kk);
        else if constexpr (B_KC) fb[buf][Fi - FN] = frag_kc(Bb, fpb[Fi - FN], kk);
        else if constexpr (ROW512) fb[buf][Fi - FN] = frag_tr_swz_rb<TILE_M * 2>(Bb, wm * WM + (Fi - FN) * 32, lane, kk);
        else fb[buf][Fi - FN] = frag_bf16<B_KC, TILE_M>(Bb, wm * WM + (Fi - FN) * 32 + l31, kk, hi);
      };
      for_seq<0, FN + FM>([&](auto f) __attribute__((always_inline)) { read_frag(0, 0, f); });
      __builtin_amdgcn_sched_barrier(0);
      // 64 MFMAs; behind MFMA q of sub-step kk: fragment q of sub-step kk + 1 (q < 8); piece P behind MFMA slot 4 P + 1 (16
      // pieces) or (2 P + 1) * 8 / 5 (20 pieces: slots 1, 4, 8, 11, 14, ... 62).
      // sched_barrier(0) after every MFMA: nothing moves across, the order below IS the instruction stream (left to itself the
      // scheduler put all 16 ds_writes - behind one s_waitcnt vmcnt(0) - and all 16 loads at the top of the step)
      for_seq<0, FN * FM * (BK / 16)>([&](auto stag) __attribute__((always_inline)) {
        constexpr int S = decltype(stag)::value, kk = S / (FN * FM), q = S % (FN * FM), i = q / FM, j = q % FM, cur = kk & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
        if constexpr (kk + 1 < BK / 16 && q < FN + FM) read_frag(kk + 1, cur ^ 1, ActTag<q>{});
        if constexpr (NP == 16) {
          if constexpr (q % 4 == 1) move_piece(ActTag<kk * 4 + q / 4>{});
        } else {
          for_seq<0, NP>([&](auto pt) __attribute__((always_inline)) {
            if constexpr ((2 * decltype(pt)::value + 1) * NSLOT / (2 * NP) == S) move_piece(pt);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      lds_barrier();
    };
    for (int h = 0; h + 1 < nk; h += 2) {
      step(h, ActTag<0>{});
      step(h + 1, ActTag<1>{});
    }
    if (nk & 1) step(nk - 1, ActTag<0>{});
  } else {
  // NS register stages of BK reduce-elements each are in flight (issue-early / write-late): for the K = 256..512
  // projections of this model EVERY operand byte of the tile is requested before the first MFMA, so a wave pays
  // about one HBM/L2 round trip for its whole main loop instead of one per K tile (measured: 2.9 K cycles per
  // K tile with a single stage, the MFMAs themselves need ~0.5 K).
  // The 64 x 64 tile runs when the grid is too small to fill the chip with big tiles (the recipe batch of 3750 frames:
  // 472 workgroups, ~2 per CU): nothing hides a workgroup's own round trips there, one stage in flight cost 0.6 us per
  // K tile (17 - 20 us for K = 2048).  Its 16 accumulator registers leave room for 4 stages (64 registers).
  constexpr int NS = A_KC ? ((TILE_N == 64 && TILE_M == 64) ? kNsSmall : (TILE_M > 128 ? 1 : kNsKc))
                          : (TILE_M > 128 ? 1 : 2);              // (two stages of a 128x256 tile pair would spill)
  uint4 ra[NS][TILE_N / 32], rb[NS][TILE_M / 32];
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
#pragma unroll
    for (int i = 0; i < TILE_N / 32; ++i) ra[s_][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TILE_M / 32; ++i) rb[s_][i] = make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int kk = kbeg + s_ * BK;
    if (kk < kend && !ab_nold) {
      load_a(ra[s_], kk);
      load_b(rb[s_], kk);
    }
  }
  SMX_STAMP(1);
  for (int kb = kbeg; kb < kend; kb += NS * BK) {
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int k0 = kb + s_ * BK;
      if (k0 >= kend) break;
      if constexpr (!A_KC) {
        if (do_cs) stage_colsum<T, TILE_N>(ra[s_], cs);
      }
      stage_store<T, A_KC, TILE_N>(ra[s_], As, t);
      stage_store<T, B_KC, TILE_M>(rb[s_], Bs, t);
      lds_barrier();
      if (k0 + NS * BK < kend && !ab_nold) {
        load_a(ra[s_], k0 + NS * BK);
        load_b(rb[s_], k0 + NS * BK);
      }
      if (ab_nomfma) {
      } else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
          bf16x8 fa[FN], fb[FM];
#pragma unroll
          for (int i = 0; i < FN; ++i) {
            if constexpr (A_KC) fa[i] = frag_kc(As, fpa[i], kk);
            else fa[i] = frag_bf16<A_KC, TILE_N>(As, wn * WN + i * 32 + l31, kk, hi);
          }
#pragma unroll
          for (int j = 0; j < FM; ++j) {
            if constexpr (B_KC) fb[j] = frag_kc(Bs, fpb[j], kk);
            else fb[j] = frag_bf16<B_KC, TILE_M>(Bs, wm * WM + j * 32 + l31, kk, hi);
          }
#pragma unroll
          for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
      } else {
        const float* Af = reinterpret_cast<const float*>(As);
        const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll 4
        for (int s2 = 0; s2 < BK / 2; ++s2) {
          int k = 2 * s2 + hi;
          float fa[FN], fb[FM];
#pragma unroll
          for (int i = 0; i < FN; ++i) fa[i] = Af[k * (TILE_N + 4) + wn * WN + i * 32 + l31];
#pragma unroll
          for (int j = 0; j < FM; ++j) fb[j] = Bf[k * (TILE_M + 4) + wm * WM + j * 32 + l31];
#pragma unroll
          for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
      }
      lds_barrier();
    }
  }

  }   // (register-staged main loop)
  if constexpr (!A_KC) {
    if (do_cs) {                                          // (uniform per workgroup)
      constexpr int CPK = TILE_N / CSN;                   // column chunks per k row of the stage = threads per group
      float* redc = reinterpret_cast<float*>(smem);       // [256 / CPK groups][TILE_N]
      lds_barrier();                                      // every wave is done reading As / Bs
#pragma unroll
      for (int q = 0; q < CSN; ++q) redc[(t / CPK) * TILE_N + (t % CPK) * CSN + q] = cs[q];
      lds_barrier();
      if (t < TILE_N && n0 + t < p.N) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 256 / CPK; ++g) sum += redc[g * TILE_N + t];
        p.acolsum[((long)split * p.batch + bz) * p.N + n0 + t] = sum;
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  if (ab_nost) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    if (s == 123.456f) reinterpret_cast<float*>(p.C)[0] = s;
    return;
  }
  SMX_STAMP(2);
  const smx_epilogue& e = p.e;
  if (e.out_mode == SMX_OUT_ATOMIC_F32) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WN + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < FM; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + wm * WM + j * 32 + g * 8 + hi * 4;
          float* Cf = reinterpret_cast<float*>(p.C) + (long)bz * p.sC + (long)n * p.ldc + m;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (n < p.N && m + q < p.M) atomicAdd(Cf + q, e.alpha * acc[i][j][g * 4 + q]);
        }
    }
    return;
  }
  constexpr int STG_LD = TILE_M * 4 + 16;               // bytes per staged fp32 row (16 B pad: conflict-free b128)
  const int osz = (e.out_mode == SMX_OUT_T) ? (int)sizeof(T) : 4;
  if constexpr (ALIAS_SIDE) lds_barrier();               // every wave is done reading the operand ring
#pragma unroll
  for (int i = 0; i < NSIDE; ++i)
    if (t + 256 * i < TILE_M + TILE_N) side[t + 256 * i] = side_value(i);   // (visible after the first barrier below)
  if constexpr (TILE_M > 128) {
  // (wide tile: 128 accumulator registers - duplicating the loop per variant spills there; the choice stays inside)
  // LayerNorm fused into the epilogue (LNF; the tile holds whole rows: M == TILE_M == 256 or 512): gamma / beta are parked in
  // LDS behind the side vector (requested before the first store of the epilogue, read back without any vmcnt wait)
  float* lng = side + TILE_M + TILE_N;
  constexpr bool LNB = (LNF & 3) == 1 || (LNF & 3) == 3;  // (LNF & 4: the LayerNorm input is float32, SMX_IO_LNX_F32)
  float dgam[LNB ? 8 : 1], dbet[LNB ? 8 : 1];
  if constexpr (LNF != 0) {
#pragma unroll
    for (int cc = t; cc < TILE_M; cc += 256) {
      lng[cc] = (LNB ? e.ln_gamma : e.lnf_gamma)[cc];
      if (LNF == 2 || ((LNF & 3) == 3 && e.lnf_act != SMX_ACT_NONE)) lng[TILE_M + cc] = e.lnf_beta[cc];
    }
    if constexpr (LNB) {
#pragma unroll
      for (int q = 0; q < 8; ++q) dgam[q] = dbet[q] = 0.f;
    }
  }
  // LayerNorm forward on the float32 stream: the one-pass phase of gemm_common.h (epilogue_phase_ln1p), both row-complete tiles
  constexpr bool LN512F = LNF == 2;
  const bool ln512f = LN512F && osz == 4 && p.epi_simple == 2 && (e.io_flags & SMX_IO_RES_F32) != 0 && e.res != nullptr &&
                      !(e.flags & SMX_EPI_ACT_GRAD) && p.batch == 1;
  if constexpr (ROW512 && LNB) {
    // LayerNorm backward on the 128 x 512 tile: its own loop with the first half's side inputs requested ahead of the dump
    // (gemm_common.h).  NOT on the 128 x 256 tile: its 128 accumulators live in the arch VGPRs of a 256-register budget, the request
    // set spills 28-60 registers there (C2b step, same box: 17.74 -> 20.0 ms) - it keeps the generic phase below.
    constexpr bool EXT_ = (LNF & 3) == 3, XF_ = (LNF & 4) != 0;
    // ONE request set: the first half of a phase is requested before the accumulator dump of that phase (behind the previous
    // phase's last stores: the round trip runs under the two barriers and the dump), the second half where it is consumed.
    // (Measured, 64 000 x 2048 -> 512: the generic pair of loads per half 174.8 us; two request sets, every half prefetched: 194.8 us
    //  - 2 x 60 registers next to the 256 accumulators spill 27-150 registers into the item loops; the four phases as straight-line
    //  code, so that dumped accumulator fragments die: 86-159 spilled.)
    LnBwdIn<T, EXT_, XF_, TILE_M> in;
    ln_bwd_request<T, EXT_, XF_, TILE_M>(p, n0, t, in);
#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      lds_barrier();
      if (wn == (ph * 32) / WN) {                          // the wave row that owns these accumulator rows (128 x 512: every wave)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          if (i == ((ph * 32) % WN) / 32) {
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(smem + l31 * STG_LD + (wm * WM + j * 32 + g * 8 + hi * 4) * 4) =
                    make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          }
        }
      }
      lds_barrier();
      if (ph < 2) SMX_STAMP(3 + 2 * ph);
      ln_bwd_half<T, EXT_, XF_, TILE_M>(p, smem, lng, n0 + ph * 32, t, in, dgam, dbet);
      ln_bwd_request<T, EXT_, XF_, TILE_M>(p, n0 + ph * 32 + 16, t, in);
      ln_bwd_half<T, EXT_, XF_, TILE_M>(p, smem + 16 * STG_LD, lng, n0 + ph * 32 + 16, t, in, dgam, dbet);
      if (ph + 1 < NPH) ln_bwd_request<T, EXT_, XF_, TILE_M>(p, n0 + ph * 32 + 32, t, in);      // (uniform)
      if (ph < 2) SMX_STAMP(4 + 2 * ph);
    }
  } else {
  if constexpr (LN512F) {
    if (ln512f) {
      // (its own loop, chosen ONCE: with both paths inside one loop their hoisted row pointers and constants are live together and
      //  hipcc spills - every reload then sits behind the stores of the previous item, s_waitcnt vmcnt(0))
      float* lnst = lng + 2 * TILE_M;                      // [2][128][2]: statistics of the LayerNorm and of the optional second one
      float* lng2 = lnst + 4 * 128;                        // gamma | beta of the second LayerNorm
      if (e.lnf2_y) {                                      // (uniform; visible behind the first barrier of the loop)
#pragma unroll
        for (int cc = t; cc < TILE_M; cc += 256) { lng2[cc] = e.lnf2_gamma[cc]; lng2[TILE_M + cc] = e.lnf2_beta[cc]; }
      }
      uint32_t resw[TILE_M / 64][8];
      // the statistics block of this tile is written once, at the very end: touch its page NOW, so that the address translation
      // of that last store is not what the workgroup's slot waits for (measured: 18 us of a 209 us launch)
      float st_touch = 0.f;
      if (e.lnf_stats && t == 0) st_touch = __builtin_nontemporal_load(e.lnf_stats + 2 * (long)n0);
#pragma unroll 1
      for (int ph = 0; ph < NPH; ++ph) {
        ln1p_request_res<TILE_M>(p, n0 + ph * 32, t, resw);   // (in flight under the dump and its two barriers)
        lds_barrier();
        if (wn == (ph * 32) / WN) {                        // the wave row that owns these accumulator rows (128 x 512: every wave)
#pragma unroll
          for (int i = 0; i < FN; ++i) {
            if (i == ((ph * 32) % WN) / 32) {
#pragma unroll
              for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                  *reinterpret_cast<float4*>(smem + l31 * STG_LD + (wm * WM + j * 32 + g * 8 + hi * 4) * 4) =
                      make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
            }
          }
        }
        lds_barrier();
        if (ph < 2) SMX_STAMP(3 + 2 * ph);
        epilogue_phase_ln1p<T, TILE_M>(p, smem, side, lng, lnst, lng2, ph, n0 + ph * 32, t, resw);
        if (ph < 2) SMX_STAMP(4 + 2 * ph);
      }
      lds_barrier();
      asm volatile("" :: "v"(st_touch));
      ln1p_store_stats(p, lnst, n0, t, TILE_N);
      SMX_STAMP(7);
      return;
    }
  }
#pragma unroll 1
  for (int ph = 0; ph < NPH; ++ph) {
    const int row_in_tile = ph * PH_ROWS;
    lds_barrier();
    if (wn == row_in_tile / WN) {                         // the wave row that owns these accumulator rows
      const int i0 = (row_in_tile % WN) / 32;              // first 32-row fragment of the phase
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        if (PH_FRAGS == FN || (i >= i0 && i < i0 + PH_FRAGS)) {
#pragma unroll
          for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(smem + ((i - (PH_FRAGS == FN ? 0 : i0)) * 32 + l31) * STG_LD +
                                         (wm * WM + j * 32 + g * 8 + hi * 4) * 4) =
                  make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        }
      }
    }
    lds_barrier();
    if (ph < 2) SMX_STAMP(3 + 2 * ph);
    if constexpr (LNB) {                                  // the LayerNorm backward replaces the ordinary epilogue
      epilogue_phase_lnbwd<T, (LNF & 3) == 3, (LNF & 4) != 0, TILE_M>(p, smem, lng, n0 + row_in_tile, t, dgam, dbet);
      continue;
    }
    if (sizeof(T) == 2 && osz == 2) {
      if (VEC && p.epi_simple == 1) epilogue_phase<T, 2, TILE_N, TILE_M, VEC, 1>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
      else if (VEC && p.epi_simple == 2) epilogue_phase<T, 2, TILE_N, TILE_M, VEC, 2>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
      else epilogue_phase<T, 2, TILE_N, TILE_M, VEC>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
    } else if (VEC && p.epi_simple == 1) epilogue_phase<T, 4, TILE_N, TILE_M, VEC, 1>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
    else if (VEC && p.epi_simple == 2 && sizeof(T) == 2) epilogue_phase<T, 4, TILE_N, TILE_M, VEC, 2>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
    else epilogue_phase<T, 4, TILE_N, TILE_M, VEC>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
    if (e.colsum) {
      // column sums of this phase's outputs (the bias gradient of a fused backward): every item was written back to
      // its staged slot; thread t adds columns t, t + 256, ... over the valid rows in a fixed order
      lds_barrier();
#pragma unroll
      for (int cc = t; cc < TILE_M; cc += 256) {
        const int rows = min(PH_ROWS, p.N - (n0 + row_in_tile));
        float s = ph == 0 ? 0.f : red[cc];
        for (int r = 0; r < rows; ++r) s += *reinterpret_cast<const float*>(smem + r * STG_LD + cc * 4);
        if (ph < NPH - 1) red[cc] = s;
        else if (m0 + cc < p.M) reinterpret_cast<float*>(e.workspace)[(long)tile_n * p.M + m0 + cc] = s;
      }
    }
    if constexpr (LNF == 2) {
      // (fp32 output: the items of epilogue_phase are 4 columns wide, a thread re-reads slots other threads wrote back)
      if (osz == 4) lds_barrier();
      epilogue_phase_lnfwd<T, TILE_M>(p, smem, lng, n0 + row_in_tile, t);
    }
    if (ph < 2) SMX_STAMP(4 + 2 * ph);
  }
  }   // (generic phase loop)
  if constexpr (LNB) {
    {
      // dgamma / dbeta of the tile: the row groups (8 of 32 lanes for 256 columns, 4 of 64 lanes for 512) are folded through
      // LDS in a fixed order into ONE partial row pair per tile
      constexpr int LPR = TILE_M / 8, NG = 256 / LPR;
      float* redg = reinterpret_cast<float*>(smem);
      lds_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        redg[(t / LPR) * TILE_M + (t % LPR) * 8 + q] = dgam[q];
        redg[2048 + (t / LPR) * TILE_M + (t % LPR) * 8 + q] = dbet[q];
      }
      lds_barrier();
#pragma unroll
      for (int cc = t; cc < TILE_M; cc += 256) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) { sg += redg[g * TILE_M + cc]; sb += redg[2048 + g * TILE_M + cc]; }
        e.ln_partial[((long)tile_n * 2) * TILE_M + cc] = sg;
        e.ln_partial[((long)tile_n * 2 + 1) * TILE_M + cc] = sb;
      }
    }
  }
  } else {
  // The epilogue variant is chosen ONCE, outside the phase loop: with the choice inside, the loop carried the hoisted
  // invariants (output / side-input row pointers, masks) of all the instantiations at the same time and spilled ~90 B
  // per thread to scratch at the 168-register budget - 90 MB of extra HBM writes per launch of an output-bound kernel
  // (PMC WRITE_SIZE 350 MB for 262 MB of output).
  auto run_phases = [&](auto osz_tag, auto lvl_tag) {
    constexpr int OSZ_ = decltype(osz_tag)::value, LVL_ = decltype(lvl_tag)::value;
#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      lds_barrier();
      const int row_in_tile = ph * PH_ROWS;
      if (wn == row_in_tile / WN) {                         // the wave row that owns these accumulator rows
        const int i0 = (row_in_tile % WN) / 32;              // first 32-row fragment of the phase
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          if (PH_FRAGS == FN || (i >= i0 && i < i0 + PH_FRAGS)) {
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(smem + ((i - (PH_FRAGS == FN ? 0 : i0)) * 32 + l31) * STG_LD +
                                           (wm * WM + j * 32 + g * 8 + hi * 4) * 4) =
                    make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          }
        }
      }
      lds_barrier();
      if (ph < 2) SMX_STAMP(3 + 2 * ph);
      epilogue_phase<T, OSZ_, TILE_N, TILE_M, VEC, LVL_>(p, smem, side, ph, n0 + row_in_tile, m0, bz, split, t);
      if (LVL_ == 0 && e.colsum) {
        // column sums of this phase's outputs (the bias gradient of a fused backward): every item was written back to
        // its staged slot; thread t < TILE_M adds column t over the valid rows in a fixed order
        lds_barrier();
        if (t < TILE_M) {
          const int rows = min(PH_ROWS, p.N - (n0 + row_in_tile));
          float s = ph == 0 ? 0.f : red[t];
          for (int r = 0; r < rows; ++r) s += *reinterpret_cast<const float*>(smem + r * STG_LD + t * 4);
          if (ph < NPH - 1) red[t] = s;
          else if (m0 + t < p.M) reinterpret_cast<float*>(e.workspace)[(long)tile_n * p.M + m0 + t] = s;
        }
      }
      if (ph < 2) SMX_STAMP(4 + 2 * ph);
    }
  };
  const int lvl = VEC ? p.epi_simple : 0;
  if (sizeof(T) == 2 && osz == 2) {
    if (lvl == 1) run_phases(ActTag<2>{}, ActTag<1>{});
    else if (lvl == 2) run_phases(ActTag<2>{}, ActTag<2>{});
    else run_phases(ActTag<2>{}, ActTag<0>{});
  } else if (lvl == 1) run_phases(ActTag<4>{}, ActTag<1>{});
  else if (lvl == 2 && sizeof(T) == 2) run_phases(ActTag<4>{}, ActTag<2>{});
  else run_phases(ActTag<4>{}, ActTag<0>{});
  }
  SMX_STAMP(7);
#undef SMX_STAMP
}

}  // namespace smx
