// wgrad_group.hip — ALL the weight gradients of one encoder layer in ONE launch (bf16, gfx950).
//
//   for every weight w of the group:   dW_w (M_w x K_w) = dZ_w^T X_w     reduce over the same `rows` frames
//
// Why a grouped kernel (profiles/r01_step_c2b_v14.txt): one launch per weight needs ~24 split-K
// slices of a 1024 x 256 weight to fill 256 CUs - 42 K steps per workgroup between a DMA cold start and a slab
// epilogue, 25 MB of fp32 slabs per weight (200 MB per layer, written and read back), and with 128 x 128 tiles every
// operand byte crosses L2 -> LDS 3.2 times.  A Conformer layer has 8 such weights = 23 tiles of 256 x 256: together
// they fill the chip with 11 slices each, i.e. ~90 K steps per workgroup, 8x less slab volume, and the 256 x 256 tile
// halves the L2 -> LDS traffic per flop.
//
// Kernel: one workgroup of 512 threads (8 waves, 4 x 2) per CU owns one (weight, 256 x 256 tile, K slice) item.  Both
// operands are reduce-strided (a k row = 256 contiguous columns = 512 B) and go HBM/L2 -> LDS by
// global_load_lds_dwordx4 into a ring of NST stages of BK frames (128 KB of LDS: 2 x 64 or 4 x 32 frames), one barrier
// per K step, no operand VGPRs; fragments come out through ds_read_b64_tr_b16 (hardware 4 x 16 transpose).
// LDS image of an operand stage: element (k, c) at k * 512 + (((c >> 3) ^ ((k & 3) << 2)) << 4) + (c & 7) * 2 - the DMA
// lands a 1 KB piece (2 k rows) linearly, so the XOR is applied to the SOURCE column granule (cf. gemm_tn_dma_kernel).
// Each wave accumulates 64 x 128 outputs (8 fragments of 32 x 32, 128 accumulator registers); the bias gradient
// (column sums of dZ) is summed on the VALU from the A fragments the wave holds anyway, in the waves that own the
// first X column tile.  Output: fp32 slabs [splits][M][K] (+ [splits][M] bias partials) per weight, folded into the
// gradients by smx_reduce_jobs in a fixed order (bit-reproducible, no atomics).
#include <stdlib.h>

#include "gemm_common.h"

namespace smx {

constexpr int WG_MAX_ITEMS = SMX_WGRAD_GROUP_MAX;
constexpr int WG_TILE = 256;

struct WgItem {
  const bf16_t* A;      // dZ (rows x M), row stride lda
  const bf16_t* B;      // X  (rows x K), row stride ldb
  float* ws;            // [splits][M*K] slabs, then [splits][M] bias partials
  long lda, ldb;
  int M, K;
  int tile0, tiles_m;   // first global tile index of this weight, K / 256
  int want_bias, pad;
};

struct WgGroupParams {
  WgItem it[WG_MAX_ITEMS];
  int nitems, total_tiles, splits, ksteps_per_split, rows;   // rows: the multiple of 64 the DMA ring walks
  int tail;             // 0..63 frames behind `rows`: one or two zero-filled steps of the LAST split's workgroups
  int ablate;           // debug (env SMX_WGROUP_ABLATE): 1 = no MFMA / fragment reads, 2 = no DMA, 4 = DMA never waited for, 8 = no X pieces
  long long* dbg;       // debug (smx_debug_set_timing_buffer): per workgroup [total cycles, cycles in wait+barrier, realtime ticks, niter]
};

// (round 4: the `nt` hint on this load - operand rows streamed past the caches so that the slabs survive for smx_reduce_jobs - was
//  measured: C2b step -0.08 ms, C2a +1.0 ms (its tiles re-read the row panels through L2) and some small shapes 20x slower: not taken)
__device__ __forceinline__ void wg_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// fragment of a reduce-strided stage (512-byte k rows): lane (l31, hi) receives k = kk*16 + hi*8 .. +7 of column cbase+l31
__device__ __forceinline__ bf16x8 wg_frag(const char* lds, int cbase, int l31, int hi, int kk) {
  typedef short short4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const int lane = l31 | (hi << 5);
  const int li = lane & 15, g1 = (lane >> 4) & 1;
  const int k = kk * 16 + hi * 8 + (li >> 2);           // (k & 3) == li >> 2; row k + 4 has the same swizzle
  const int c = cbase + g1 * 16 + (li & 3) * 4;
  const char* p0 = lds + k * 512 + ((((c >> 3) ^ ((li >> 2) << 2)) << 4) + (c & 7) * 2);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * 512));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return __builtin_bit_cast(bf16x8, make_uint4(ua.x, ua.y, ub.x, ub.y));
}

// acc += sum of the 8 bf16 of a fragment: four v_dot2c_f32_bf16 against (1, 1) (hipcc has no builtin for it on gfx950;
// the shift / mask / add formulation was ~40 VALU instructions per fragment pair and 16-frame sub-step - more issue
// slots than the 8 MFMAs next to them leave free)
__device__ __forceinline__ void wg_sum8(float& acc0, float& acc1, const bf16x8& f) {
  const uint4 u = __builtin_bit_cast(uint4, f);
  const uint32_t one2 = 0x3F803F80u;
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc0) : "v"(u.x), "v"(one2));
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc1) : "v"(u.y), "v"(one2));
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc0) : "v"(u.z), "v"(one2));
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc1) : "v"(u.w), "v"(one2));
}

template <int BK>
__global__ __launch_bounds__(512, 2) void wgrad_group_kernel(const WgGroupParams p) {
  constexpr int NST = 128 / BK;                          // ring stages in 128 KB
  constexpr int OP_BYTES = BK * 512, STAGE_BYTES = 2 * OP_BYTES;
  constexpr int NPC = BK / 16;                           // 1 KB pieces (2 k rows) per wave, operand and stage
  static_assert(BK == 64 || BK == 32, "ring of 2 x 64 or 4 x 32 frames");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave >> 1, wm = wave & 1, l31 = lane & 31, hi = lane >> 5;

  // ---- work item: each XCD (workgroup id % 8) takes a contiguous run of the (split, tile) list, so the tiles of one
  // weight that read the same K range sit on the same XCD's L2 at the same time
  const int nwork = p.total_tiles * p.splits;
  const int per = (nwork + 7) >> 3;
  const int q = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || q >= nwork) return;
  const int split = q / p.total_tiles, tg = q % p.total_tiles;
  int w = 0;
#pragma unroll 1
  for (int i = 1; i < p.nitems; ++i)
    if (tg >= p.it[i].tile0) w = i;
  const WgItem& it = p.it[w];
  const int tl = tg - it.tile0, tile_n = tl / it.tiles_m, tile_m = tl % it.tiles_m;
  const int n0 = tile_n * WG_TILE, m0 = tile_m * WG_TILE;
  const int kbeg = split * p.ksteps_per_split * 64;
  const int kend = min(p.rows, kbeg + p.ksteps_per_split * 64);
  const int niter = kend > kbeg ? (kend - kbeg) / BK : 0;

  // ---- DMA source pointers: piece pc = wave + 8 j holds k rows 2 pc, 2 pc + 1 of the stage; lane -> (row, granule)
  const int prow = lane >> 5, krow = 2 * wave + prow;    // (k row index mod 4 is the same for every j: 16 j = 0 mod 4)
  const int gsrc = ((lane & 31) ^ ((krow & 3) << 2)) * 8;
  const bf16_t* pa[NPC];
  const bf16_t* pb[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const long kr = kbeg + 2 * (wave + 8 * j) + prow;
    pa[j] = it.A + kr * it.lda + n0 + gsrc;
    pb[j] = it.B + kr * it.ldb + m0 + gsrc;
  }
  const long stepa = (long)BK * it.lda, stepb = (long)BK * it.ldb;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);
#ifdef SMX_DIAG
  const bool ab_nomfma = p.ablate & 1, ab_nodma = p.ablate & 2;
#else
  constexpr bool ab_nomfma = false, ab_nodma = false;     // (ablations compiled out of the product)
#endif
  // pieces j0 .. j0 + nj - 1 of stage s (both operands)
  auto issue_part = [&](int s, auto j0_tag, auto nj_tag) {
    constexpr int J0 = decltype(j0_tag)::value, NJ = decltype(nj_tag)::value;
    if (ab_nodma) return;
    const uint32_t dst = wave_lds + (s % NST) * STAGE_BYTES;
#pragma unroll
    for (int j = J0; j < J0 + NJ; ++j) {
      wg_glds16(pa[j], dst + j * 8192);
#ifdef SMX_DIAG
      if (!(p.ablate & 8)) wg_glds16(pb[j], dst + OP_BYTES + j * 8192);      // (8: no B pieces - is the DMA rate the bound?)
#else
      wg_glds16(pb[j], dst + OP_BYTES + j * 8192);
#endif
      pa[j] += stepa;
      pb[j] += stepb;
    }
  };
  auto issue = [&](int s) { issue_part(s, ActTag<0>{}, ActTag<NPC>{}); };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const bool do_cs = it.want_bias && tile_m == 0 && wm == 0;     // (uniform per wave)

#ifdef SMX_DIAG
  const bool ab_nowait = p.ablate & 4;
#else
  constexpr bool ab_nowait = false;
#endif
#ifdef SMX_DIAG   // (clock stamps for tools/wgroup_stamps.py: diagnostic build only)
  long long t_start = 0, t_wait = 0, r_start = 0;
  if (p.dbg) { t_start = clock64(); r_start = wall_clock64(); }
#endif
  for (int s = 0; s < NST - 1 && s < niter; ++s) issue(s);
  for (int itn = 0; itn < niter; ++itn) {
#ifdef SMX_DIAG
    long long tw0 = 0;
    if (p.dbg) tw0 = clock64();
#endif
    // this wave's pieces of stage `itn` have landed once at most the (2 NPC each) DMA instructions of the younger stages
    // in flight are outstanding (vmcnt retires in order; nothing else uses vector memory in this loop)
    const int ahead = min(NST - 2, niter - 1 - itn);
    if (ab_nowait) {}
    else if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPC) : "memory");
    else if (NST >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                                        // ... and everybody's; the stage read last step is free again
    // the refill of the stage that was read last step is issued piecewise between the MFMA groups below (a DMA piece
    // costs 100-200 issue cycles: eight of them in front of the first MFMA left the matrix pipe idle after every barrier)
#ifdef SMX_DIAG
    if (p.dbg) t_wait += clock64() - tw0;
#endif
    const bool refill = itn + NST - 1 < niter;
    const char* As = smem + (itn % NST) * STAGE_BYTES;
    const char* Bs = As + OP_BYTES;
    if (ab_nomfma) { if (refill) issue(itn + NST - 1); continue; }
    // ping-pong: waves w and w + 4 share a SIMD; the upper four issue the whole refill BEFORE their MFMAs, the lower four
    // AFTER theirs, so that on every SIMD one wave's DMA issue runs beside the other wave's matrix work
    if (wave >= 4) {
      if (refill) issue(itn + NST - 1);
    }
    bf16x8 fa[2][2], fb[2][4];                           // fragments double-buffered over the 16-frame sub-steps
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = wg_frag(As, wn * 64 + i * 32, l31, hi, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = wg_frag(Bs, wm * 128 + j * 32, l31, hi, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[nxt][i] = wg_frag(As, wn * 64 + i * 32, l31, hi, kk + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[nxt][j] = wg_frag(Bs, wm * 128 + j * 32, l31, hi, kk + 1);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wg_sum8(bsum[i][0], bsum[i][1], fa[cur][i]);
      }
    }
    if (wave < 4) {
      if (refill) issue(itn + NST - 1);
    }
  }

  // ---- ragged tail (rows % 64 frames): the workgroups of the last split stage them with guarded 16-byte loads - zeros
  // behind the last frame - into the image the DMA would have written, and multiply them like any other stage.  (Peeling the
  // tail on the host cost one slab GEMM + one reduction per weight: 96 + 96 launches = 12 % of the recipe batch's step for
  // 38 of its 3750 frames.)
  if (p.tail > 0 && split == p.splits - 1) {
#pragma unroll 1
    for (int ts = 0; ts * BK < p.tail; ++ts) {
      wg_barrier();                                        // every wave is done with the stage it read last
      const int kb = p.rows + ts * BK, kmax = p.rows + p.tail;
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int kr = kb + 2 * (wave + 8 * j) + prow;
        uint4 va = make_uint4(0, 0, 0, 0), vb = va;
        if (kr < kmax) {
          va = *reinterpret_cast<const uint4*>(it.A + (long)kr * it.lda + n0 + gsrc);
          vb = *reinterpret_cast<const uint4*>(it.B + (long)kr * it.ldb + m0 + gsrc);
        }
        char* d = smem + wave * 1024 + j * 8192 + lane * 16;
        *reinterpret_cast<uint4*>(d) = va;
        *reinterpret_cast<uint4*>(d + OP_BYTES) = vb;
      }
      __syncthreads();
      const char* As = smem;
      const char* Bs = As + OP_BYTES;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 fa[2], fb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = wg_frag(As, wn * 64 + i * 32, l31, hi, kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = wg_frag(Bs, wm * 128 + j * 32, l31, hi, kk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (do_cs) {
#pragma unroll
          for (int i = 0; i < 2; ++i) wg_sum8(bsum[i][0], bsum[i][1], fa[i]);
        }
      }
    }
  }

#ifdef SMX_DIAG
  if (p.dbg && t == 0) {
    long long* d = p.dbg + (long)blockIdx.x * 4;
    d[0] = clock64() - t_start; d[1] = t_wait; d[2] = wall_clock64() - r_start; d[3] = niter;
  }
#endif
  // ---- slab: acc[i][j][g*4 + q] is dW[n0 + wn*64 + i*32 + l31][m0 + wm*128 + j*32 + g*8 + hi*4 + q] ----------------
  float* slab = it.ws + (long)split * it.M * it.K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn * 64 + i * 32 + l31;
    float* row = slab + (long)n * it.K + m0 + wm * 128 + hi * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + j * 32 + g * 8) =
            make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
  }
  if (do_cs) {
    float* bpart = it.ws + (long)p.splits * it.M * it.K + (long)split * it.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float b = bsum[i][0] + bsum[i][1];
      const float s = b + __shfl_xor(b, 32, 64);                   // the two k halves (hi = 0 / 1) of the same column
      if (hi == 0) bpart[n0 + wn * 64 + i * 32 + l31] = s;
    }
  }
}

// =================================================================================================================================
// Round 6, small batches: the weight gradients of a layer in ONE launch WITHOUT split-K slabs.  A few thousand frames are a short
// reduction: the 256 x 256 kernel above needs 2 K-slices per tile to occupy 184 of 256 CUs (recipe batch, 3750 frames: 65 us), writes
// 47 MB of float32 slabs and leaves smx_reduce_jobs 100 MB to fold (25-33 us).  Here every 128 x 128 tile of every weight walks ALL
// the frames (368 tiles at d_model 512: two workgroups per CU, one round) on the LDS-DMA ring of gemm_tn_dma_kernel (gemm.hip: both
// operands reduce-strided, 64-frame stages, ring of two, no operand VGPRs) and adds its tile INTO the gradient (float32
// read-modify-write, each element owned by exactly one workgroup: fixed order, bit-reproducible, no atomics, no workspace).
// Bias gradient: one extra MFMA per fragment against a ones fragment in the waves of the first X column tile, added into dbias.
// =================================================================================================================================
struct WgDirectItem {
  const bf16_t* A; const bf16_t* B;     // dZ (rows x M), X (rows x K)
  float* dW; float* dbias;              // (M x K) float32, row stride lddw; (M) or null
  long lda, ldb, lddw;
  int M, K, tile0, tiles_m;             // first global tile of this weight, K / 128
};
struct WgDirectParams {
  WgDirectItem it[WG_MAX_ITEMS];
  int nitems, total_tiles, rows;
};

#ifndef SMX_WGD_BK
#define SMX_WGD_BK 64     // frames per ring stage of the slab-free grouped wgrad: 64 = ring of two, 32 = ring of four (the same 64 KB)
#endif
__global__ __launch_bounds__(256, 2) void wgrad_group_direct_kernel(const WgDirectParams p) {
  typedef bf16_t T;
  constexpr int BK = SMX_WGD_BK, NST = 128 / BK, TILE = 128, WN = 64, WM = 64, FN = 2, FM = 2;
  constexpr int OP_BYTES = BK * TILE * 2, STAGE_BYTES = 2 * OP_BYTES, NPC = BK / 16;
  constexpr int PH_ROWS = 64, STG_LD = TILE * 4 + 16;
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE_BYTES];      // the ring (64 KB); the epilogue rows alias it
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wm = wave & 1, l31 = lane & 31, hi = lane >> 5;
  // each XCD (workgroup id % 8) takes a contiguous run of tiles: the tiles of one weight that share operand columns meet in one L2
  const int per = (p.total_tiles + 7) >> 3;
  const int tg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || tg >= p.total_tiles) return;
  int w = 0;
#pragma unroll 1
  for (int i = 1; i < p.nitems; ++i)
    if (tg >= p.it[i].tile0) w = i;
  const WgDirectItem& it = p.it[w];
  const int tl = tg - it.tile0, tile_n = tl / it.tiles_m, tile_m = tl % it.tiles_m;
  const int n0 = tile_n * TILE, m0 = tile_m * TILE;
  const int rows_main = p.rows - p.rows % BK, niter = rows_main / BK;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const int prow = lane >> 4, gsrc = ((lane & 15) ^ (prow << 2)) * 8;      // this lane's k row in a piece, source column
  const T* pa[NPC];
  const T* pb[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const long kr = 4 * (wave + 4 * j) + prow;
    pa[j] = it.A + kr * it.lda + n0 + gsrc;
    pb[j] = it.B + kr * it.ldb + m0 + gsrc;
  }
  const long stepa = (long)BK * it.lda, stepb = (long)BK * it.ldb;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  auto issue = [&](int itn) {
    const uint32_t dst = wave_lds + (itn % NST) * STAGE_BYTES;
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
  const bool do_cs = it.dbias != nullptr && tile_m == 0 && wm == 0;          // (uniform per wave)
  const uint32_t one2 = 0x3F803F80u;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(one2, one2, one2, one2));
  auto multiply = [&](const char* As, const char* Bs) __attribute__((always_inline)) {
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
  };
  for (int s_ = 0; s_ < NST - 1 && s_ < niter; ++s_) issue(s_);
  for (int itn = 0; itn < niter; ++itn) {
    // this wave's pieces of stage itn have landed when at most the (2 NPC each) DMA instructions of the younger stages in flight are
    // outstanding (vmcnt retires in order; nothing else uses vector memory in this loop)
    const int ahead = min(NST - 2, niter - 1 - itn);
    if (ahead >= 2 && NST >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPC) : "memory");
    else if (ahead >= 1 && NST >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                       // ... and everybody's; the stage read last step is free again
    if (itn + NST - 1 < niter) issue(itn + NST - 1);
    const char* As = smem + (itn % NST) * STAGE_BYTES;
    multiply(As, As + OP_BYTES);
  }
  // ---- ragged tail (rows % 64 frames): guarded 16-byte loads - zeros behind the last frame - into the image the DMA writes ----
  if (p.rows > rows_main) {
    lds_barrier();                                       // every wave is done with the stage it read last
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int kr = rows_main + 4 * (wave + 4 * j) + prow;
      uint4 va = make_uint4(0, 0, 0, 0), vb = va;
      if (kr < p.rows) {
        va = *reinterpret_cast<const uint4*>(it.A + (long)kr * it.lda + n0 + gsrc);
        vb = *reinterpret_cast<const uint4*>(it.B + (long)kr * it.ldb + m0 + gsrc);
      }
      char* d = smem + wave * 1024 + j * 4096 + lane * 16;
      *reinterpret_cast<uint4*>(d) = va;
      *reinterpret_cast<uint4*>(d + OP_BYTES) = vb;
    }
    __syncthreads();
    multiply(smem, smem + OP_BYTES);
  }
  if (do_cs && hi == 0) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WN + i * 32 + l31;
      it.dbias[n] += accb[i][0];                         // (this wave is the only writer of these 32 entries)
    }
  }
  // ---- epilogue: 64 staged rows at a time, dW += tile (float4 read-modify-write, whole 512-byte row segments) ----
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
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
    // thread t: column group (t & 31) * 4, rows (t >> 5) + 8 k
    float* gp = it.dW + (long)(n0 + ph * PH_ROWS + (t >> 5)) * it.lddw + m0 + (t & 31) * 4;
    float4 old[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) old[k] = *reinterpret_cast<const float4*>(gp + (long)(8 * k) * it.lddw);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(smem + ((t >> 5) + 8 * k) * STG_LD + (t & 31) * 16);
      *reinterpret_cast<float4*>(gp + (long)(8 * k) * it.lddw) = make_float4(old[k].x + a.x, old[k].y + a.y, old[k].z + a.z, old[k].w + a.w);
    }
  }
}

static int wg_splits(int rows, int total_tiles) {
  // K slices per tile: tiles * s workgroups run in rounds of one per CU, so the choice is a quantisation problem: 23 tiles
  // (C2b layer) x 11 = 253 fills one round; 92 tiles (d_model 512) x 2 = 184 leaves 28 % of the chip idle, x 3 = 276 needs a
  // second round for 20 workgroups, x 8 = 736 fills 2.9 rounds (C2a step 49.4 -> 48.7 ms).  Every extra slice costs one more
  // fp32 slab per tile (written, read back by smx_reduce_jobs): score = fill of the last round - 1 % per slice.
  if (total_tiles < 1) total_tiles = 1;
  const int nk = rows / 64;
  const int smax = nk / 8 > 0 ? nk / 8 : 1;                        // at least 8 K steps (512 frames) per slice
  // Cost model (microseconds; fitted to the rocprof averages of the four bench configs): a round of up to 256 workgroups
  // walks its 64-frame steps at ~2 us each, and every slice adds one fp32 slab per tile that is written here and read back by
  // smx_reduce_jobs (2 x 256 KB at ~4 TB/s).  At 64 000 frames the slab term is 1-2 % per slice (the "fill - 1 % per slice" rule
  // of round 2 picked the same counts: C2b 11, C2a 8); at the recipe's 3750 frames it is 17 % per slice, and five slices
  // (72 + 32 us) lose to two (measured below).
  const int cus = 256;
  const double t_step = 2.0, t_slab = 0.131;
  int best = 1;
  double best_t = 1e30;
  for (int s = 1; s <= smax && (long)s * total_tiles <= 4L * cus; ++s) {
    const long w = (long)s * total_tiles;
    const double rounds = (double)((w + cus - 1) / cus);
    const double t = rounds * (double)((nk + s - 1) / s) * t_step + (double)s * total_tiles * t_slab;
    if (t < best_t - 1e-9) { best_t = t; best = s; }
  }
  return best;
}

}  // namespace smx

using namespace smx;

#ifdef SMX_DIAG   // diagnostic build only: the product library has no global mutable state and no debug export
static long long* g_wg_dbg = nullptr;
extern "C" void smx_debug_set_wgroup_timing_buffer(void* p) { g_wg_dbg = reinterpret_cast<long long*>(p); }
#endif

extern "C" int smx_wgrad_group_splits(int rows, const smx_wgrad_item* items, int nitems) {
  if (!items || nitems <= 0 || rows < 64) return 0;
  int tiles = 0;
  for (int i = 0; i < nitems; ++i) tiles += (items[i].M / WG_TILE) * (items[i].K / WG_TILE);
  return wg_splits(rows - rows % 64, tiles);
}

extern "C" size_t smx_wgrad_group_workspace(int M, int K, int splits) {
  if (M <= 0 || K <= 0 || splits <= 0) return 0;
  return (size_t)splits * ((size_t)M * K + M) * sizeof(float);
}

extern "C" int smx_wgrad_group(int dtype, int rows, const smx_wgrad_item* items, int nitems, int splits, void* stream) {
  SMX_REQUIRE(dtype == SMX_BF16, "smx_wgrad_group: bf16 only (fp32 weights take smx_linear_wgrad)");
  SMX_REQUIRE(items && nitems >= 1 && nitems <= WG_MAX_ITEMS, "smx_wgrad_group: 1..%d items per launch", WG_MAX_ITEMS);
  SMX_REQUIRE(rows >= 64 && splits >= 1, "smx_wgrad_group: at least 64 rows");
  WgGroupParams p;
  memset(&p, 0, sizeof(p));
  int tiles = 0;
  for (int i = 0; i < nitems; ++i) {
    const smx_wgrad_item& s = items[i];
    SMX_REQUIRE(s.dZ && s.X && s.workspace, "smx_wgrad_group: null pointer in item %d", i);
    SMX_REQUIRE(s.M > 0 && s.K > 0 && s.M % WG_TILE == 0 && s.K % WG_TILE == 0, "smx_wgrad_group: item %d: M, K must be multiples of 256", i);
    SMX_REQUIRE(aligned16(s.dZ) && aligned16(s.X) && aligned16(s.workspace) && s.lddz % 8 == 0 && s.ldx % 8 == 0 &&
                s.lddz >= s.M && s.ldx >= s.K, "smx_wgrad_group: item %d: operands must be 16-byte aligned with ld %% 8 == 0", i);
    WgItem& d = p.it[i];
    d.A = reinterpret_cast<const bf16_t*>(s.dZ); d.B = reinterpret_cast<const bf16_t*>(s.X);
    d.ws = reinterpret_cast<float*>(s.workspace);
    d.lda = s.lddz; d.ldb = s.ldx; d.M = s.M; d.K = s.K;
    d.tile0 = tiles; d.tiles_m = s.K / WG_TILE; d.want_bias = s.want_bias;
    tiles += (s.M / WG_TILE) * (s.K / WG_TILE);
  }
  p.nitems = nitems; p.total_tiles = tiles; p.splits = splits; p.rows = rows - rows % 64; p.tail = rows % 64;
  const int nk = rows / 64;
  SMX_REQUIRE(splits <= nk, "smx_wgrad_group: more splits than 64-frame steps");
  p.ksteps_per_split = (nk + splits - 1) / splits;
  const int nwork = tiles * splits, per = (nwork + 7) / 8;
  const int ablate_env = cfg().wgroup_ablate;             // (0 unless built with -DSMX_DIAG)
  p.ablate = ablate_env;
#ifdef SMX_DIAG
  p.dbg = g_wg_dbg;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return fail(SMX_ELAUNCH, "smx_wgrad_group: cannot reserve 128 KB of LDS");
    attr_done = true;
  }
  // 32-frame ring stages (ring of four) with the ping-pong refill order: the measured best of {32, 64} x {none, ping-pong, + mid-step barrier}
  hipLaunchKernelGGL(wgrad_group_kernel<32>, dim3(8 * per), dim3(512), 131072, s, p);
  return check_launch("smx_wgrad_group");
}

extern "C" int smx_wgrad_group_direct_ok(int rows, int M, int K) {
  return rows >= 1 && M > 0 && K > 0 && M % 128 == 0 && K % 128 == 0;
}

extern "C" int smx_wgrad_group_direct(int dtype, int rows, const smx_wgrad_direct_item* items, int nitems, void* stream) {
  SMX_REQUIRE(dtype == SMX_BF16, "smx_wgrad_group_direct: bf16 only");
  SMX_REQUIRE(items && nitems >= 1 && nitems <= WG_MAX_ITEMS, "smx_wgrad_group_direct: 1..%d items per launch", WG_MAX_ITEMS);
  SMX_REQUIRE(rows >= 1, "smx_wgrad_group_direct: no rows");
  WgDirectParams p;
  memset(&p, 0, sizeof(p));
  int tiles = 0;
  for (int i = 0; i < nitems; ++i) {
    const smx_wgrad_direct_item& s = items[i];
    SMX_REQUIRE(s.dZ && s.X && s.dW, "smx_wgrad_group_direct: null pointer in item %d", i);
    SMX_REQUIRE(smx_wgrad_group_direct_ok(rows, s.M, s.K), "smx_wgrad_group_direct: item %d: M, K must be multiples of 128", i);
    SMX_REQUIRE(aligned16(s.dZ) && aligned16(s.X) && aligned16(s.dW) && s.lddz % 8 == 0 && s.ldx % 8 == 0 && s.lddw % 4 == 0 &&
                    s.lddz >= s.M && s.ldx >= s.K && s.lddw >= s.K,
                "smx_wgrad_group_direct: item %d: operands must be 16-byte aligned (lddz, ldx %% 8 == 0, lddw %% 4 == 0)", i);
    WgDirectItem& d = p.it[i];
    d.A = reinterpret_cast<const bf16_t*>(s.dZ); d.B = reinterpret_cast<const bf16_t*>(s.X);
    d.dW = s.dW; d.dbias = s.dbias;
    d.lda = s.lddz; d.ldb = s.ldx; d.lddw = s.lddw; d.M = s.M; d.K = s.K;
    d.tile0 = tiles; d.tiles_m = s.K / 128;
    tiles += (s.M / 128) * (s.K / 128);
  }
  p.nitems = nitems; p.total_tiles = tiles; p.rows = rows;
  const int per = (tiles + 7) / 8;
  hipLaunchKernelGGL(wgrad_group_direct_kernel, dim3(8 * per), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("smx_wgrad_group_direct");
}
