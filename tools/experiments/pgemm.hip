// pgemm.hip — persistent LDS-DMA GEMM for the long-K bf16 NT / NN shapes of the encoder (gfx950).
//
// Why (DESIGN_APPENDIX.md §4/§5): the tiled kernels of gemm.hip run 2-3 workgroups of 4 waves per CU; each fetches its operands
// through registers (one 48 KB stage in flight per workgroup), multiplies, stages its epilogue and retires.  On the
// K >= 512 shapes that structure is bound by (operand bytes in flight per CU) / (L2 -> CU latency): a 128 x 256 tile
// needs 96 B of operands per output-tile row per K element, and prefetching across the epilogue needs VGPRs it does not
// have.  First attempt here (256 x 128 tile, 64-element stages, ring of 2): no faster - one 48 KB stage in flight and
// half the flops per operand byte of this version.
//
// Here ONE workgroup of 512 threads (8 waves, 4 x 2, 64 x 128 outputs each) per CU walks a list of 256 x 256 output
// tiles: twice the flops per operand byte.  Both operands go HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPRs) into a
// ring of THREE 32 KB stages of 32 reduce elements (two stages = 64 KB in flight while one is multiplied), and the stage
// stream does not stop at a tile boundary: the first three stages of the next tile are requested before / while the
// epilogue of the current tile runs from its own 33 KB of staging rows.  The epilogue is the shared epilogue_phase
// (gemm_common.h): bias, C0 side inputs, saved pre-activation, activation / activation gradient, dropout, row mask,
// residual, alpha.
//
// LDS images: a reduce-contiguous operand stage is [256 rows][32 k] = 64-byte rows, 16-byte chunk c of row r at chunk
// position c ^ ((r >> 2) & 3) (any 16 rows x one chunk cover the 64 banks once: conflict-free ds_read_b128 fragments);
// the DMA lands a 1 KB piece (16 rows) linearly, so the XOR is applied to the SOURCE chunk.  A reduce-strided stage
// (dgrad weights, [32 k][256 columns] = 512-byte k rows) uses the image of wgrad_group.hip (ds_read_b64_tr_b16).
//
// vmcnt bookkeeping (hipcc does not count the asm DMA; vmcnt retires in order and counts stores): a wave's wait for a
// stage names how many YOUNGER vector-memory instructions may stay outstanding: the DMA pieces of the stages issued
// after it, and - for the first stages of a tile - a LOWER bound of the store instructions the previous epilogue
// certainly issued (a smaller count only waits longer).  So the stores of tile i drain under the MFMAs of tile i + 1.
#include <stdlib.h>

// (included at the end of summarymixing_amd/csrc/gemm.hip under -DSMX_PGEMM_BUILD: gemm_common.h is already in)

namespace smx {

constexpr int PG_TN = 256, PG_TM = 256, PG_BK = 32, PG_NTHR = 512;
constexpr int PG_A_BYTES = PG_TN * PG_BK * 2;            // 256 rows x 32 k x 2 B = 16 KB
constexpr int PG_B_BYTES = PG_TM * PG_BK * 2;            // 256 rows x 32 k (NT) or 32 k rows x 256 columns (NN) = 16 KB
constexpr int PG_STAGE = PG_A_BYTES + PG_B_BYTES;        // 32 KB
constexpr int PG_NST = 3;
constexpr int PG_PHR = 32, PG_NPH = PG_TN / PG_PHR;      // epilogue: 8 phases of 32 staged rows
constexpr int PG_STG_LD = PG_TM * 4 + 16;
constexpr int PG_EPI_BYTES = PG_PHR * PG_STG_LD;         // 33 KB of fp32 staging rows, outside the ring
constexpr int PG_SIDE_OFF = PG_NST * PG_STAGE + PG_EPI_BYTES;
constexpr int PG_LDS = PG_SIDE_OFF + (PG_TM + PG_TN) * 4;   // + bias[256] | row factors[256]
constexpr int PG_NPA = 2, PG_NPB = 2;                    // 1 KB DMA pieces per wave and stage (16 + 16 pieces, 8 waves)
constexpr int PG_NP = PG_NPA + PG_NPB;

// wait until at most n younger vector-memory instructions are outstanding (n rounded DOWN to a multiple of 4, <= 60:
// a smaller count is always safe)
__device__ __forceinline__ void pg_wait_n(int n) {
#define PG_W(N) case N / 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); break;
  switch (min(n, 60) >> 2) {
    PG_W(0) PG_W(4) PG_W(8) PG_W(12) PG_W(16) PG_W(20) PG_W(24) PG_W(28) PG_W(32) PG_W(36) PG_W(40) PG_W(44) PG_W(48) PG_W(52) PG_W(56) PG_W(60)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef PG_W
}

// fragment of a reduce-contiguous stage (64-byte rows): k = kk*16 + hi*8 .. +7 of row r
__device__ __forceinline__ uint32_t pg_pre(int r, int hi) { return (uint32_t)(r * 64 + ((hi ^ ((r >> 2) & 3)) << 4)); }
__device__ __forceinline__ bf16x8 pg_frag_kc(const char* lds, uint32_t pre, int kk) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (pre ^ (uint32_t)(kk << 5))));
}
// fragment of a reduce-strided stage with 512-byte k rows (256 columns), granule XOR 4 * (k & 3) (wgrad_group.hip)
__device__ __forceinline__ bf16x8 pg_frag_ks(const char* lds, int cbase, int l31, int hi, int kk) {
  typedef short short4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const int lane = l31 | (hi << 5);
  const int li = lane & 15, g1 = (lane >> 4) & 1;
  const int k = kk * 16 + hi * 8 + (li >> 2);
  const int c = cbase + g1 * 16 + (li & 3) * 4;
  const char* p0 = lds + k * 512 + ((((c >> 3) ^ ((li >> 2) << 2)) << 4) + (c & 7) * 2);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * 512));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return __builtin_bit_cast(bf16x8, make_uint4(ua.x, ua.y, ub.x, ub.y));
}

template <bool B_KC>
__global__ __launch_bounds__(PG_NTHR, 2) void pgemm_kernel(GemmParams p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* stg = smem + PG_NST * PG_STAGE;
  float* side = reinterpret_cast<float*>(smem + PG_SIDE_OFF);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave >> 1, wm = wave & 1, l31 = lane & 31, hi = lane >> 5;

  // ---- tile list: XCD x (workgroup id % 8) owns a contiguous run of tile ids (m fastest), its workgroups take them
  // round robin, so the M tiles of one A row panel are in flight together on ONE L2
  const int ntiles = p.tiles_n * p.tiles_m;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int t_hi = (int)((long)ntiles * (xcd + 1) / 8);
  int cur = (int)((long)ntiles * xcd / 8) + li;
  if (cur >= t_hi) return;
  const int nk = p.K / PG_BK;
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);

  // ---- the DMA issue stream (runs ahead of the compute position, across tile boundaries) ----
  const T* pa[PG_NPA];
  const T* pb[PG_NPB];
  const long stepb = B_KC ? PG_BK : (long)PG_BK * p.ldb;
  int iss_tile = cur, iss_ks = 0, issued = 0;            // next stage to issue; stages issued so far
  auto set_issue_tile = [&](int tile) {
    const int n0i = (tile / p.tiles_m) * PG_TN, m0i = (tile % p.tiles_m) * PG_TM;
#pragma unroll
    for (int j = 0; j < PG_NPA; ++j) {                   // piece = 16 rows x 64 B; lane -> (row, 16-byte chunk position)
      const int row = (wave + 8 * j) * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);   // chunk position (lane & 3) of the row holds source chunk `chunk`
      pa[j] = A + (long)min(n0i + row, p.N - 1) * p.lda + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < PG_NPB; ++j) {
      if constexpr (B_KC) {
        const int row = (wave + 8 * j) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        pb[j] = B + (long)min(m0i + row, p.M - 1) * p.ldb + chunk * 8;
      } else {                                           // piece = 2 k rows x 512 B; granule XOR 4 * (k & 3)
        const int krow = (wave + 8 * j) * 2 + (lane >> 5);
        pb[j] = B + (long)krow * p.ldb + m0i + (((lane & 31) ^ ((krow & 3) << 2)) * 8);
      }
    }
  };
  auto issue_next = [&]() {                              // stage `issued` -> slot issued % NST
    const uint32_t dst = wave_lds + (issued % PG_NST) * PG_STAGE;
#pragma unroll
    for (int j = 0; j < PG_NPA; ++j) { glds16(pa[j], dst + j * 8192); pa[j] += PG_BK; }
#pragma unroll
    for (int j = 0; j < PG_NPB; ++j) { glds16(pb[j], dst + PG_A_BYTES + j * 8192); pb[j] += stepb; }
    ++issued;
    if (++iss_ks == nk) {
      iss_ks = 0;
      iss_tile += nloc;
      if (iss_tile < t_hi) set_issue_tile(iss_tile);
    }
  };
  set_issue_tile(cur);
  issue_next();
  issue_next();                                          // (nk >= NST: both belong to the first tile)

  uint32_t fpa[2], fpb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) fpa[i] = pg_pre(wn * 64 + i * 32 + l31, hi);
#pragma unroll
  for (int j = 0; j < 4; ++j) fpb[j] = pg_pre(wm * 128 + j * 32 + l31, hi);

  const smx_epilogue& e = p.e;
  const int osz = (e.out_mode == SMX_OUT_T) ? 2 : 4;
  const int nit = osz == 2 ? 2 : 4;                       // items per thread and phase (epilogue_phase: NIT)
  const int stores_full = PG_NPH * nit * (1 + ((e.z && !(e.flags & SMX_EPI_ACT_GRAD)) ? 1 : 0));
  int done = 0;                                           // stages consumed so far
  int prev_stores = 0;                                    // lower bound of the previous epilogue's store instructions
  int store_steps = 0;                                    // K steps of this tile whose awaited stage is OLDER than those stores

  for (; cur < t_hi; cur += nloc) {
    const int tile_n = cur / p.tiles_m, tile_m = cur % p.tiles_m;
    const int n0 = tile_n * PG_TN, m0 = tile_m * PG_TM;
    // epilogue side vector of this thread: t < 256 -> bias[m0 + t], then 256 row factors row_mask[n] * alpha
    float side_v = 0.f;
    if (t < PG_TM) {
      if (e.bias && m0 + t < p.M) side_v = e.bias[m0 + t];
    } else {
      const int n = n0 + t - PG_TM;
      side_v = ((e.row_mask && n < p.N) ? (e.row_mask[n] ? 1.f : 0.f) : 1.f) * e.alpha;
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    for (int ks = 0; ks < nk; ++ks) {
      // ---- stage `done` has landed: younger = the DMA pieces of the stages issued after it (+ the previous epilogue's
      // stores while the awaited stage is one that was issued before them) ----
      pg_wait_n((issued - done - 1) * PG_NP + (ks < store_steps ? prev_stores : 0));
      lds_barrier();                                     // everybody's pieces; the slot read last step is free again
      if (ks == PG_NST) {                                // (by now older than every awaited stage: it has arrived)
        settle(side_v);
        side[t] = side_v;
      }
      // ping-pong (p.reg_epi reused as the knob in this experiment build: SMX_PGEMM_PP): waves w and w + 4 share a SIMD; the
      // upper four issue their DMA pieces BEFORE their MFMAs, the lower four AFTER, so that on every SIMD one wave's
      // vector-memory issue (~100-180 cycles per piece) runs beside the other wave's matrix work
      const bool can_issue = issued - done < PG_NST && iss_tile < t_hi;
      const bool pp = p.reg_epi != 0;
      if (can_issue && (!pp || wave >= 4)) issue_next();
      const char* As = smem + (done % PG_NST) * PG_STAGE;
      const char* Bs = As + PG_A_BYTES;
#pragma unroll
      for (int kk = 0; kk < PG_BK / 16; ++kk) {
        bf16x8 fa[2], fb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = pg_frag_kc(As, fpa[i], kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (B_KC) fb[j] = pg_frag_kc(Bs, fpb[j], kk);
          else fb[j] = pg_frag_ks(Bs, wm * 128 + j * 32, l31, hi, kk);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
      if (can_issue && pp && wave < 4) issue_next();
      ++done;
    }

    // ---- epilogue of this tile; the ring keeps filling with the next tile's first stages ----
    lds_barrier();                                       // every wave is done with the last stage: its slot is free
    const int before = issued - done;                    // stages of the next tile already in flight (issued before the stores)
    if (issued - done < PG_NST && iss_tile < t_hi) issue_next();
    store_steps = issued - done;                         // ... and these are the K steps that wait for such a stage
    (void)before;
    auto run_phases = [&](auto osz_tag, auto lvl_tag) {
      constexpr int OSZ_ = decltype(osz_tag)::value, LVL_ = decltype(lvl_tag)::value;
#pragma unroll 1
      for (int ph2 = 0; ph2 < PG_NPH / 2; ++ph2) {       // wave row ph2 owns rows ph2 * 64 .. + 63: two phases of 32 rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // (unrolled: the accumulator fragment index stays a constant)
          const int ph = ph2 * 2 + i;
          if (ph) lds_barrier();                         // the previous phase's rows have been read
          if (wn == ph2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stg + l31 * PG_STG_LD + (wm * 128 + j * 32 + g * 8 + hi * 4) * 4) =
                    make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          }
          lds_barrier();
          epilogue_phase<T, OSZ_, PG_TN, PG_TM, true, LVL_, PG_PHR, PG_NTHR>(p, stg, side, ph, n0 + ph * PG_PHR, m0, 0, 0, t);
        }
      }
    };
    const int lvl = p.epi_simple;
    if (osz == 2) {
      if (lvl == 1) run_phases(ActTag<2>{}, ActTag<1>{});
      else if (lvl == 2) run_phases(ActTag<2>{}, ActTag<2>{});
      else run_phases(ActTag<2>{}, ActTag<0>{});
    } else if (lvl == 1) run_phases(ActTag<4>{}, ActTag<1>{});
    else run_phases(ActTag<4>{}, ActTag<0>{});
    prev_stores = (n0 + PG_TN <= p.N) ? stores_full : 0;  // (a partial last row tile skips stores: no lower bound)
  }
}

static int pg_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
    cus -= cus % 8;
  }
  return cus;
}

// Shapes the persistent kernel takes (the caller has checked bf16 + 16-byte aligned "vec" operands and epilogue):
// long reductions with an output width that is a multiple of 256, and enough 256 x 256 tiles to give most CUs one.
bool pgemm_eligible(const GemmParams& p, bool b_kc) {
  static const int env = getenv("SMX_PGEMM") ? atoi(getenv("SMX_PGEMM")) : 1;
  static const int kmin = getenv("SMX_PGEMM_KMIN") ? atoi(getenv("SMX_PGEMM_KMIN")) : 512;
  if (!env || p.batch != 1 || p.splits != 1 || p.K % PG_BK != 0 || p.K < kmin || p.K < (PG_NST + 1) * PG_BK || p.M % PG_TM != 0 ||
      p.N < PG_TN)
    return false;
  if (p.e.out_mode == SMX_OUT_ATOMIC_F32 || p.e.colsum || p.ablate || !p.epi_lds) return false;
  (void)b_kc;
  const long ntiles = (long)((p.N + PG_TN - 1) / PG_TN) * (p.M / PG_TM);
  return ntiles * 10 >= pg_cus() * 9L;                    // (about) one tile per workgroup at least
}

int launch_pgemm(GemmParams& p, bool b_kc, hipStream_t s) {
  static const int pp_env = getenv("SMX_PGEMM_PP") ? atoi(getenv("SMX_PGEMM_PP")) : 1;
  p.reg_epi = pp_env;                                     // (the persistent kernel has no register-domain epilogue: field reused)
  p.tiles_n = (p.N + PG_TN - 1) / PG_TN;
  p.tiles_m = p.M / PG_TM;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS) != hipSuccess)
      return fail(SMX_ELAUNCH, "smx_gemm: cannot reserve %d bytes of LDS for the persistent kernel", PG_LDS);
    attr_done = true;
  }
  if (b_kc) hipLaunchKernelGGL(pgemm_kernel<true>, dim3(pg_cus()), dim3(PG_NTHR), PG_LDS, s, p);
  else hipLaunchKernelGGL(pgemm_kernel<false>, dim3(pg_cus()), dim3(PG_NTHR), PG_LDS, s, p);
  return check_launch("smx_gemm");
}

}  // namespace smx
