// gemm_panel.h — the panel-resident GEMM of libsmx.so (gfx950) for the SHORT reductions with WIDE outputs of the encoder layers:
// the FFN up-projection  Z = X W1^T + b1, H = D(act(Z))  (Conformer.py:458-472, K = d_model, M = d_ffn) and the act-grad dgrad
// of the down-projection  dZ = D(dY W2 * act'(Z))  (its autograd backward).  Both are OUTPUT bound (K <= 512: 227 flop/B at
// d_model = 512, 120 at 256), and on the tiled kernels of gemm_kernel.h the LDS is what saturates first: per 128 x 128 tile and
// K = 512 they move 256 KB of operands VGPR -> LDS (at ~80 B/clk/CU) and read 512 KB of fragments for 4100 cycles of MFMA issue.
//
// Structure (one 512-thread workgroup = 8 waves per CU and per 128-row panel of the activations):
//  * the PANEL, 128 rows x K of A, is staged in LDS once (128 KB at K = 512; XOR-swizzled 16-byte chunks, conflict-free
//    ds_read_b128 fragments) and read by every wave for every column chunk: A crosses L2 -> LDS once per 128 x M outputs;
//  * the WEIGHTS never touch LDS: they are PRE-PACKED in MFMA fragment order (smx_weight_pack: the 1 KB fragment of 32 columns x
//    16 reduce elements is contiguous, lane-major), so a wave's B fragment is ONE perfectly coalesced buffer_load_dwordx4 with a
//    scalar offset - straight into the registers the MFMA reads, through a ring of 8 steps in flight.  The packing kernel also
//    absorbs the transposition the dgrad needs (W2 is (d_model, d_ffn): reduce-strided), once per weight and step;
//  * every wave owns 128 rows x 64 columns at a time (128 accumulator registers) and walks its own column chunks (wave w:
//    chunks w, w + 8, ...) with NO workgroup barrier after the panel load: two waves share a SIMD, so one wave's epilogue
//    (VALU: activation, dropout hash, conversions; stores) runs under the other wave's MFMAs;
//  * the bias rides in the packed image as one more fragment per column block (reduce slots 0 / 1 = the bf16 high and low parts
//    of the float32 bias, against a constant activation fragment of ones: exact to 2^-17): no bias load in the kernel at all;
//  * epilogue per 32-row block: accumulators -> bf16 -> a wave-private 4 KB
//    LDS scratch (transposition only: ds_write_b64 in, ds_read_b128 out, no barrier) -> whole 128-byte row segments per store.
//    The activation is evaluated on the bf16-rounded pre-activation (what torch autocast does: the Linear's output IS bf16);
//    the act-grad form requests its 16 saved pre-activation items into the B ring's registers during the last 8 steps of the
//    main loop (ahead of the chunk's first store: vmcnt retires in order and counts stores).
// Dropout: the same counter-based mask as everywhere (dropout_apply<8> on n * M + m), bit-identical to smx_gemm's.
#pragma once
#include "gemm_common.h"

namespace smx {

struct PanelParams {
  const bf16_t* A; long lda;          // (N, K) activations, reduce-contiguous
  const void* Bp;                     // packed weight: [M / 32][K / 16 + 1][64 lanes][8 bf16] (the last fragment of a block: the bias)
  bf16_t* C; long ldc;                // (N, M) output
  bf16_t* Z; long ldz;                // MODE 0: saved pre-activation (output, may be null); MODE 1: the saved pre-activation (input)
  int N, M;
  unsigned dthresh; float dscale; uint64_t seed; const uint64_t* epoch;
  int nt;                             // 2: stream C past the caches
  int drop_cols;                      // dropout on the first drop_cols output columns only (a multiple of 64), mask index n * drop_cols + m
  const uint8_t* row_mask; float alpha;   // per-row keep mask [N] or null; output scale (folded into the dropout scale)
  int rows;                           // rows per panel: 128, 64 or 32 (the kernel's ROWS)
  // MODE 2 (split-K slabs, round 6): workgroup (x, y) multiplies K-slice y - columns [y K, (y + 1) K) of A, packed image y of Bp -
  // and stores its float32 partial products to slab y: no bias, no activation, no mask (smx_slab_epilogue applies them to the sum)
  float* slab; long slab_stride;      // [nslice][N][M] float32, elements between slabs
  int nslice;
  long b_slice_bytes;                 // bytes between the packed images of consecutive K-slices
  int csplit;                         // workgroups per panel: workgroup (panel, s) takes the chunk rounds s, s + csplit, ... (small N: fill the chip)
  long long* dbg;                     // -DSMX_DIAG only: per-wave clock stamps (tools/panel_stamps.py)
};

typedef uint32_t pg_u32x4 __attribute__((ext_vector_type(4)));
#ifndef SMX_PANEL_ABL        // experiment builds only (tools/experiments/panel_variant.sh): 1 = no epilogue, 2 = no MFMA, 4 = no weight loads, 8 = no fragment reads
#define SMX_PANEL_ABL 0
#endif

// ---- weight / pre-activation loads.  Every chunk requests its own ring at its head, BEHIND the previous chunk's stores (vmcnt
// retires in order and counts stores, so the first fragments wait for that write drain; the partner wave of the SIMD runs
// meanwhile).  Two ways around that drain were built and measured at no gain (+-2 %, inside the box-to-box spread): requesting
// the next chunk's ring in the tail of the main loop needs either loads the compiler does not count (inline asm + hand-placed
// s_waitcnt - the register allocator then copies the asm outputs before the data has arrived: nondeterministic garbage at
// full-chip sizes) or dummy stores that make both predecessors of the loop header look alike to hipcc's waitcnt merge (spills
// the ring at K = 256).
__device__ __forceinline__ void panel_ld(uint4& dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff) {
  if constexpr ((SMX_PANEL_ABL & 4) != 0) { asm volatile("" : "+v"(dst.x), "+v"(dst.y), "+v"(dst.z), "+v"(dst.w)); return; }
  const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  dst = make_uint4(r.x, r.y, r.z, r.w);
}

// act(v) * s with the scale folded into the activation's own arithmetic (s = the inverted-dropout scale, inv_s = 1 / s; both 1
// without dropout): Swish x / ((1 + e^-x) / s) costs what the unscaled form costs, GELU folds s into its 0.5
template <int ACT>
__device__ __forceinline__ float panel_act_scaled(float v, float s, float inv_s) {
  if constexpr (ACT == SMX_ACT_SWISH) {
    return v * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(-1.4426950408889634f * v), inv_s, inv_s));
  } else if constexpr (ACT == SMX_ACT_GELU) {
    float e;
    const float hs = 0.5f * s;
    return v * fmaf(gelu_parts(v, e), hs, hs);
  } else if constexpr (ACT == SMX_ACT_RELU) {
    return fmaxf(v, 0.f) * s;
  } else {
    return v * s;
  }
}
// The keep decisions of dropout_apply<8> (smx_common.h) for 8 consecutive elements whose first index is 2 * p0 (p0 a multiple of
// 4, pair indices below 2^32: N * M < 2^30 here), bit for bit: the seed / high-word mix `hm0` is a kernel constant, and
// (p0 + q) ^ hm = (p0 ^ hm) ^ q for q < 4.  Survivors are NOT scaled (the caller folded the scale into the values).
__device__ __forceinline__ void panel_dropout8(float (&v)[8], uint32_t hm0, uint32_t p0, uint32_t t16) {
  const uint32_t ph = p0 ^ hm0 ^ pair_hi_mix(p0);
#pragma unroll
  for (int q2 = 0; q2 < 4; ++q2) {
    const uint32_t h = mix32_1(ph ^ (uint32_t)q2);
    v[2 * q2] = (h & 0xffffu) >= t16 ? v[2 * q2] : 0.f;
    v[2 * q2 + 1] = (h >> 16) >= t16 ? v[2 * q2 + 1] : 0.f;
  }
}

// ROWS (round 6): rows of the resident panel.  128 = the kernel as described above.  64 / 32: the SMALL-N forms - a batch of a few
// thousand frames has too few 128-row panels for 256 CUs (the recipe's 3750 frames: 30), and on the tiled kernels those launches are
// LDS-bound at one or two workgroups per CU (tools/experiments/r06_smalln/README.md); a 64-row panel gives 59 workgroups per chunk
// round, each wave a 64 x 64 (32 x 64) output chunk: the activation fragments still cross the LDS once per 64 output columns and
// the weights not at all.
template <int K, int MODE, int ACT, int ROWS = 128>
__global__ __launch_bounds__(512) void gemm_panel_kernel(PanelParams p) {
  static_assert(K == 256 || K == 512, "panel GEMM: K = 256 or 512");
  static_assert(ROWS == 128 || ROWS == 64 || ROWS == 32, "panel GEMM: 128 / 64 / 32 rows per panel");
  constexpr int RB = ROWS / 32;                         // 32-row accumulator blocks per wave
  constexpr int KS = K / 16, ROWB = K * 2, A_BYTES = ROWS * ROWB, SCR = MODE == 2 ? 8192 : 4096, PF = 8;
  static_assert(KS % PF == 0, "whole ring turns");
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + 8 * SCR];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (wave-uniform for the compiler: scalar ring offsets, no waterfall loops)
  const int n0 = (int)(blockIdx.x / (unsigned)p.csplit) * ROWS, csi = (int)(blockIdx.x % (unsigned)p.csplit);
#ifdef SMX_DIAG   // per-wave clock stamps: [0] start, [1] panel in LDS, [2 + 2 r] main loop of chunk round r done, [3 + 2 r] its epilogue
  long long* dbgp = p.dbg ? p.dbg + ((long)blockIdx.x * 8 + (t >> 6)) * 16 : nullptr;
#define SMX_PSTAMP(k) do { if (dbgp && lane == 0 && (k) < 16) dbgp[k] = clock64(); } while (0)
#else
#define SMX_PSTAMP(k) do { } while (0)
#endif
  SMX_PSTAMP(0);

  const int slice = MODE == 2 ? (int)blockIdx.y : 0;
  const __amdgpu_buffer_rsrc_t rb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.Bp)) + (long)slice * p.b_slice_bytes,
                                                                          (short)0, (int)((long)p.M * (K + 16) * 2), 0x00020000);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // ---- B ring: fragment (column block c * 2 + j, step kk) = 1 KB at ((c * 2 + j) * (KS + 1) + kk) * 1024, lane-major; step KS = the bias.  The chunk's base
  // rides in the VECTOR offset (range-checked: a chunk beyond the last one reads zeros, no memory touched), step and j are constants.
  uint4 rb[PF][2], bfrag[2];
  auto ld_b = [&](uint4& dst, uint32_t voff, int kk, int j) __attribute__((always_inline)) {
    panel_ld(dst, voff, rb_rs, (uint32_t)(j * (KS + 1) + kk) * 1024u);
  };
  auto chunk_voff = [&](int c) __attribute__((always_inline)) -> uint32_t { return lane16 + (uint32_t)c * (uint32_t)(2 * (KS + 1) * 1024); };

  // ---- the panel: 128 rows x K -> LDS, 16-byte chunk c of row r at chunk position c ^ (r & 15) ----
  {
    constexpr int CPR = K / 8, NA = ROWS * CPR / 512;
    const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (long)slice * K, (short)0,
                                                                            (int)((((long)p.N - 1) * p.lda + K) * 2), 0x00020000);
    uint4 ra[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + 512 * i, row = v / CPR, c = v % CPR, n = min(n0 + row, p.N - 1);   // (rows beyond N: row N - 1 again, their outputs are dropped)
      const uint32_t off = (uint32_t)(((long)n * p.lda + c * 8) * 2);
      const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra_rs, off, 0, 0);
      // row mask: a masked row enters the panel as zeros, so its accumulators hold nothing but the bias - which MODE 0 withholds
      // from it as well (the ones fragment below): act(0) = 0 for every activation here, act-grad and dropout keep the zero
      const bool keep = (MODE != 2 && p.row_mask) ? p.row_mask[n] != 0 : true;
      ra[i] = keep ? make_uint4(r.x, r.y, r.z, r.w) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + 512 * i, row = v / CPR, c = v % CPR;
      *reinterpret_cast<uint4*>(smem + row * ROWB + ((c ^ (row & 15)) << 4)) = ra[i];
    }
  }
  __syncthreads();
  SMX_PSTAMP(1);

  // C and Z (MODE 0: saved pre-activation out, may be null; MODE 1: in) as buffer resources: rows >= N are out of range, so the
  // padded rows of the last panel are dropped (stores) / read as zeros (loads) by the hardware
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)((((long)p.N - 1) * p.ldc + p.M) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz_rs = __builtin_amdgcn_make_buffer_rsrc(p.Z ? p.Z : p.C, (short)0, (int)((((long)p.N - 1) * (p.Z ? p.ldz : p.ldc) + p.M) * 2), 0x00020000);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u, ldz2 = (uint32_t)p.ldz * 2u;
  // fragment address of this lane in the panel: row l31 (+ 32 i), chunk (kk * 2 + hi) ^ (l31 & 15) = base ^ (kk << 5)
  const uint32_t a_base = (uint32_t)(l31 * ROWB + ((hi ^ (l31 & 15)) << 4));
  char* scr = smem + A_BYTES + wave * SCR;
  // dropout: the seed / high-word part of the hash is a kernel constant (pair indices < 2^32), the scale is folded into the values
  const uint64_t dseed = p.dthresh ? epoch_seed(p.seed, p.epoch) : 0;
  const uint32_t hm0 = mix32((uint32_t)dseed) ^ (uint32_t)(dseed >> 32), t16 = p.dthresh >> 16;
  const float dsc_d = (p.dthresh ? p.dscale : 1.f) * p.alpha, dsc_n = p.alpha;   // (the output scale alpha rides in the same factor)
  const int nch = p.M >> 6;

#pragma unroll 1
  for (int ch = wave + 8 * csi; ch < nch; ch += 8 * p.csplit) {
    // ---- this chunk's weight ring (first PF steps) and bias fragments ----
    uint32_t b_cur = chunk_voff(ch);
    asm volatile("" : "+v"(b_cur));
#pragma unroll
    for (int s = 0; s < PF; ++s) { ld_b(rb[s][0], b_cur, s, 0); ld_b(rb[s][1], b_cur, s, 1); }
    if constexpr (MODE == 0) { ld_b(bfrag[0], b_cur, KS, 0); ld_b(bfrag[1], b_cur, KS, 1); }
    // ---- accumulators: MODE 0 starts them at the bias (8 MFMAs of the packed bias fragment against the ones fragment, C = 0),
    // MODE 1 lets the first step's MFMAs write them (C = 0): no zero fill, no bias load ----
    f32x16 acc[RB][2];
    if constexpr (MODE == 0) {
      // reduce slots 0 and 1 = 1.0, the rest 0 - per 32-row block, and 0 for a masked row (its bias stays out: see the panel load);
      // rebuilt per chunk (the four mask bytes come with the ring's requests) rather than kept in registers across the loop
      const uint8_t* mkp = p.row_mask;
      asm volatile("" : "+s"(mkp));
      uint32_t zr = 0u;
      asm volatile("" : "+v"(zr));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const bool keep = !hi && (mkp ? mkp[min(n0 + i * 32 + l31, p.N - 1)] != 0 : true);
        const uint4 ones = make_uint4(keep ? 0x3f803f80u : 0u, zr, zr, zr);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfrag[j]), __builtin_bit_cast(bf16x8, ones), zero, 0, 0, 0);
      }
    }
    // this lane's byte offset of (row n0 + (lane >> 3), columns ch * 64 + (lane & 7) * 8 ..) in Z (the tail's requests need it; the
    // offsets the epilogue alone needs are built behind the main loop: two registers fewer across it)
    uint32_t z_off0 = (uint32_t)(((long)(n0 + (lane >> 3)) * p.ldz + ch * 64 + (lane & 7) * 8) * 2);
    // (opaque: else the loop-invariant part of all 16 item offsets is hoisted out of the chunk loop - 50 registers)
    asm volatile("" : "+v"(z_off0));
    // MODE 1: item q (0..15) of the chunk's saved pre-activation: rows i * 32 + pp * 8 + (lane >> 3) (q = i * 4 + pp), 8 columns
    auto ld_z = [&](uint4& dst, int q) __attribute__((always_inline)) {
      panel_ld(dst, z_off0 + (uint32_t)((q >> 2) * 32 + (q & 3) * 8) * ldz2, rz_rs, 0u);   // (rows beyond N: outside the resource, zeros)
    };

    // ---- main loop: KS steps of 16 reduce elements, 8 MFMAs each; the ring holds this chunk's first PF steps already ----
    uint4 fa[RB];
    // (opaque per chunk: otherwise every step's fragment address - loop invariant - is hoisted out of the chunk loop, 64 live registers)
    uint32_t a_cur = a_base;
    asm volatile("" : "+v"(a_cur));
#pragma unroll
    for (int i = 0; i < RB; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + a_cur + i * 32 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
    // The instruction order IS the source order (sched_barrier(0) after every MFMA; left to itself the scheduler hoists the fragment
    // reads and ring refills of many steps to the top and spills 90-150 registers).  Activation-fragment major: MFMA 2 i + j
    // multiplies activation fragment i by weight fragment j, so fragment i is free behind MFMA 2 i + 1 and its ONE register set is
    // re-read for the next step right there (7 MFMAs ahead of its next use); the ring slot is refilled behind MFMAs 6 and 7.
    // MODE 1: the last PF steps refill the ring's registers with this chunk's 16 saved pre-activation items (ahead of its stores).
    for_seq<0, KS>([&](auto ktag) __attribute__((always_inline)) {
      constexpr int kk = decltype(ktag)::value, slot = kk % PF;
      const uint32_t an = a_cur ^ (uint32_t)((kk + 1) << 5);
      for_seq<0, 2 * RB>([&](auto mtag) __attribute__((always_inline)) {
        constexpr int mm = decltype(mtag)::value, i = mm >> 1, j = mm & 1;
        if constexpr ((SMX_PANEL_ABL & 2) != 0) {
          if constexpr (MODE != 0 && kk == 0) { for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f; }
          asm volatile("" : "+v"(rb[slot][j].x), "+v"(fa[i].x), "+v"(acc[i][j]));
        } else if constexpr (MODE != 0 && kk == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot][j]), __builtin_bit_cast(bf16x8, fa[i]), zero, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot][j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
        }
        if constexpr (kk + 1 < KS && j == 1 && (SMX_PANEL_ABL & 8) == 0) fa[i] = *reinterpret_cast<const uint4*>(smem + an + i * 32 * ROWB);
        if constexpr (mm >= 2 * RB - 2) {                    // (behind the step's last two MFMAs)
          constexpr int jj = mm - (2 * RB - 2);
          if constexpr (kk + PF < KS) ld_b(rb[slot][jj], b_cur, kk + PF, jj);
          else if constexpr (MODE == 1 && 2 * slot + jj < 4 * RB) ld_z(rb[slot][jj], 2 * slot + jj);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    // ---- epilogue, 32 rows at a time through the wave's own scratch ----
    uint32_t c_off0 = (uint32_t)(((long)(n0 + (lane >> 3)) * p.ldc + ch * 64 + (lane & 7) * 8) * 2);
    uint32_t d_p0 = ((uint32_t)(n0 + (lane >> 3)) * (uint32_t)p.drop_cols + (uint32_t)(ch * 64 + (lane & 7) * 8)) >> 1;   // dropout pair index (N * M < 2^30)
    const bool dchunk = p.dthresh != 0 && ch * 64 < p.drop_cols;      // (uniform: this chunk's columns are dropped out)
    const float dsc = dchunk ? dsc_d : dsc_n, dinv = 1.f / dsc;
    asm volatile("" : "+v"(c_off0), "+v"(d_p0));
    // (scratch addresses rebuilt here, per chunk: hoisted out of the chunk loop they are 12 registers live across the main loop)
    uint32_t s_wr = (uint32_t)(l31 * 128 + hi * 8), s_x = (uint32_t)(l31 & 7), s_rd = (uint32_t)((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4));
    asm volatile("" : "+v"(s_wr), "+v"(s_x), "+v"(s_rd));
    SMX_PSTAMP(2 + 2 * (ch >> 3));
    if constexpr ((SMX_PANEL_ABL & 1) != 0) {
      float sabl = 0.f;
      for (int i = 0; i < RB; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sabl += acc[i][j][e];
      if (sabl == 123.456f) p.C[0].v = 1;
      if constexpr (MODE == 1) { for (int s = 0; s < PF; ++s) if (rb[s][0].x == 0x12345u) p.C[1].v = 1; }
      continue;
    }
    if constexpr (MODE == 2) {
      // ---- float32 slab: 32 rows x 64 columns at a time through the wave's 8 KB scratch (row r = 256 B, 16-byte granule q at
      // q ^ (r & 15): conflict-free both ways), out as whole 256-byte row segments, four rows per store instruction ----
      const __amdgpu_buffer_rsrc_t rs_rs = __builtin_amdgcn_make_buffer_rsrc(p.slab + (long)slice * p.slab_stride, (short)0,
                                                                              (int)((long)p.N * p.M * 4), 0x00020000);
      uint32_t f_wr = (uint32_t)(l31 * 256), f_x = (uint32_t)(l31 & 15);
      uint32_t f_rd = (uint32_t)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4));        // row lane >> 4 (+ 4 per pass), granule lane & 15
      uint32_t f_off = (uint32_t)(((long)(n0 + (lane >> 4)) * p.M + ch * 64 + (lane & 15) * 4) * 4);
      asm volatile("" : "+v"(f_wr), "+v"(f_x), "+v"(f_rd), "+v"(f_off));
      for_seq<0, RB>([&](auto itag) __attribute__((always_inline)) {
        constexpr int i = decltype(itag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(scr + f_wr + (((uint32_t)(j * 8 + g * 2 + hi) ^ f_x) << 4)) =
                make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
          // (row pp * 4 + (lane >> 4): its swizzle key is (row & 15) = ((pp & 3) * 4 + (lane >> 4)): fold pp's part into the address)
          const uint4 v = *reinterpret_cast<const uint4*>(scr + ((f_rd + (uint32_t)(pp * 4 * 256)) ^ (uint32_t)(((pp & 3) * 4) << 4)));
          const pg_u32x4 vu = {v.x, v.y, v.z, v.w};
          __builtin_amdgcn_raw_buffer_store_b128(vu, rs_rs, f_off + (uint32_t)((i * 32 + pp * 4) * p.M * 4), 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the scratch is rewritten by the next 32-row block)
      });
      SMX_PSTAMP(3 + 2 * (ch >> 3));
      continue;
    }
    for_seq<0, RB>([&](auto itag) __attribute__((always_inline)) {
      constexpr int i = decltype(itag)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack_bf16x2(acc[i][j][g * 4], acc[i][j][g * 4 + 1]);
          w.y = pack_bf16x2(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          *reinterpret_cast<uint2*>(scr + s_wr + (((uint32_t)(j * 4 + g) ^ s_x) << 4)) = w;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for_seq<0, 4>([&](auto ptag) __attribute__((always_inline)) {
        constexpr int pp = decltype(ptag)::value;
        constexpr int rstep = i * 32 + pp * 8;               // rows below this lane's first row (n0 + (lane >> 3))
        const uint4 zz = *reinterpret_cast<const uint4*>(scr + s_rd + pp * 8 * 128);
        float v[8];
        { const uint32_t w_[4] = {zz.x, zz.y, zz.z, zz.w}; unpack_words<bf16_t, 8>(w_, v); }
        // stores: buffer offsets = the chunk's lane offset + a wave-uniform row step; rows >= N fall outside the resource
        const uint32_t coff = c_off0 + (uint32_t)rstep * ldc2;
        if constexpr (MODE == 0) {
          if (p.Z) {
            const pg_u32x4 zu = {zz.x, zz.y, zz.z, zz.w};
            __builtin_amdgcn_raw_buffer_store_b128(zu, rz_rs, z_off0 + (uint32_t)rstep * ldz2, 0, 2);   // (nt: not read again before the backward pass)
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = panel_act_scaled<ACT>(v[q], dsc, dinv);
        } else {
          constexpr int q = i * 4 + pp;
          float zf[8];
          { const uint32_t w_[4] = {rb[q >> 1][q & 1].x, rb[q >> 1][q & 1].y, rb[q >> 1][q & 1].z, rb[q >> 1][q & 1].w}; unpack_words<bf16_t, 8>(w_, zf); }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= act_grad_c<ACT>(zf[e]) * dsc;
        }
        if (dchunk) panel_dropout8(v, hm0, d_p0 + (uint32_t)(rstep / 2) * (uint32_t)p.drop_cols, t16);
        const pg_u32x4 cu = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (p.nt & 2) __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 0);
      });
    });
    SMX_PSTAMP(3 + 2 * (ch >> 3));
  }
#undef SMX_PSTAMP
}

// host side: one launcher per mode (gemm_panel.hip: forward, gemm_panel_bwd.hip: act-grad), K and the activation by switch
int launch_panel_fwd(const PanelParams& p, int K, int act, hipStream_t s);
int launch_panel_actgrad(const PanelParams& p, int K, int act, hipStream_t s);
int launch_panel_slabs(const PanelParams& p, int K, hipStream_t s);

template <int MODE>
static int launch_panel_mode(const PanelParams& p, int K, int act, hipStream_t s) {
  const int rows = p.rows;
  const dim3 grid(((p.N + rows - 1) / rows) * p.csplit, MODE == 2 ? p.nslice : 1), block(512);
#define SMX_PANEL_CASE(KK, AA) \
  if (K == KK && act == AA) { \
    if (rows == 128) hipLaunchKernelGGL((gemm_panel_kernel<KK, MODE, AA, 128>), grid, block, 0, s, p); \
    else if (rows == 64) hipLaunchKernelGGL((gemm_panel_kernel<KK, MODE, AA, 64>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((gemm_panel_kernel<KK, MODE, AA, 32>), grid, block, 0, s, p); \
    return check_launch("smx_gemm_panel"); }
  SMX_PANEL_CASE(256, SMX_ACT_NONE) SMX_PANEL_CASE(256, SMX_ACT_SWISH) SMX_PANEL_CASE(256, SMX_ACT_GELU) SMX_PANEL_CASE(256, SMX_ACT_RELU)
  SMX_PANEL_CASE(512, SMX_ACT_NONE) SMX_PANEL_CASE(512, SMX_ACT_SWISH) SMX_PANEL_CASE(512, SMX_ACT_GELU) SMX_PANEL_CASE(512, SMX_ACT_RELU)
#undef SMX_PANEL_CASE
  return fail(SMX_EUNSUPPORTED, "smx_gemm_panel: K = %d / activation %d has no instantiation", K, act);
}

}  // namespace smx
