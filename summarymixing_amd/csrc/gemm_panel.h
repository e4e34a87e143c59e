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
//  * epilogue per 32-row block: accumulators (+ bias, which is the accumulator's INITIAL value) -> bf16 -> a wave-private 4 KB
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
  const void* Bp;                     // packed weight: [M / 32][K / 16][64 lanes][8 bf16]
  bf16_t* C; long ldc;                // (N, M) output
  bf16_t* Z; long ldz;                // MODE 0: saved pre-activation (output, may be null); MODE 1: the saved pre-activation (input)
  const float* bias;                  // MODE 0: [M] or null
  int N, M;
  unsigned dthresh; float dscale; uint64_t seed; const uint64_t* epoch;
  int nt;                             // 2: stream C past the caches
};

typedef uint32_t pg_u32x4 __attribute__((ext_vector_type(4)));

template <int K, int MODE, int ACT>
__global__ __launch_bounds__(512) void gemm_panel_kernel(PanelParams p) {
  static_assert(K == 256 || K == 512, "panel GEMM: K = 256 or 512");
  constexpr int KS = K / 16, ROWB = K * 2, A_BYTES = 128 * ROWB, SCR = 4096, PF = 8;
  static_assert(KS % PF == 0, "whole ring turns");
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + 8 * SCR];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (wave-uniform for the compiler: scalar ring offsets, no waterfall loops)
  const int n0 = blockIdx.x * 128;

  // ---- the panel: 128 rows x K -> LDS, 16-byte chunk c of row r at chunk position c ^ (r & 15) ----
  {
    constexpr int CPR = K / 8, NA = 128 * CPR / 512;
    const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0,
                                                                            (int)((((long)p.N - 1) * p.lda + K) * 2), 0x00020000);
    uint4 ra[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + 512 * i, row = v / CPR, c = v % CPR, n = n0 + row;
      const uint32_t off = n < p.N ? (uint32_t)(((long)n * p.lda + c * 8) * 2) : 0x80000000u;
      const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra_rs, off, 0, 0);
      ra[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + 512 * i, row = v / CPR, c = v % CPR;
      *reinterpret_cast<uint4*>(smem + row * ROWB + ((c ^ (row & 15)) << 4)) = ra[i];
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Bp), (short)0, (int)((long)p.M * K * 2), 0x00020000);
  // C and Z (MODE 0: saved pre-activation out, may be null; MODE 1: in) as buffer resources: rows >= N are out of range, so the
  // padded rows of the last panel are dropped (stores) / read as zeros (loads) by the hardware
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)((((long)p.N - 1) * p.ldc + p.M) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz_rs = __builtin_amdgcn_make_buffer_rsrc(p.Z ? p.Z : p.C, (short)0, (int)((((long)p.N - 1) * (p.Z ? p.ldz : p.ldc) + p.M) * 2), 0x00020000);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u, ldz2 = (uint32_t)p.ldz * 2u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // fragment address of this lane in the panel: row l31 (+ 32 i), chunk (kk * 2 + hi) ^ (l31 & 15) = base ^ (kk << 5)
  const uint32_t a_base = (uint32_t)(l31 * ROWB + ((hi ^ (l31 & 15)) << 4));
  char* scr = smem + A_BYTES + wave * SCR;
  const uint64_t dseed = p.dthresh ? epoch_seed(p.seed, p.epoch) : 0;
  const int nch = p.M >> 6;

#pragma unroll 1
  for (int ch = wave; ch < nch; ch += 8) {
    // ---- accumulators start at the bias (MODE 0) ----
    f32x16 acc[4][2];
    if (MODE == 0 && p.bias) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch * 64 + j * 32 + g * 8 + hi * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i][j][g * 4] = b4.x; acc[i][j][g * 4 + 1] = b4.y; acc[i][j][g * 4 + 2] = b4.z; acc[i][j][g * 4 + 3] = b4.w;
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    // ---- B ring: fragment (column block ch * 2 + j, step kk) = 1 KB at ((ch * 2 + j) * KS + kk) * 1024, lane-major ----
    const uint32_t bs0 = (uint32_t)(ch * 2) * (uint32_t)(KS * 1024);
    // this lane's byte offsets of (row n0 + (lane >> 3), columns ch * 64 + (lane & 7) * 8 ..) in C / Z and its dropout index there
    uint32_t c_off0 = (uint32_t)(((long)(n0 + (lane >> 3)) * p.ldc + ch * 64 + (lane & 7) * 8) * 2);
    uint32_t z_off0 = (uint32_t)(((long)(n0 + (lane >> 3)) * p.ldz + ch * 64 + (lane & 7) * 8) * 2);
    uint64_t d_idx0 = (uint64_t)(n0 + (lane >> 3)) * (uint64_t)p.M + (uint64_t)(ch * 64 + (lane & 7) * 8);
    // (opaque: else the loop-invariant part of all 16 item offsets of the epilogue is hoisted out of the chunk loop - 50 registers)
    asm volatile("" : "+v"(c_off0), "+v"(z_off0), "+v"(d_idx0));
    uint4 rb[PF][2];
    auto ld_b = [&](int kk, int j) __attribute__((always_inline)) -> uint4 {
      const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rb_rs, lane16, bs0 + (uint32_t)(j * KS + kk) * 1024u, 0);
      return make_uint4(r.x, r.y, r.z, r.w);
    };
    // MODE 1: item q (0..15) of the chunk's saved pre-activation: rows i * 32 + pp * 8 + (lane >> 3) (q = i * 4 + pp), 8 columns
    auto ld_z = [&](int q) __attribute__((always_inline)) -> uint4 {
      const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rz_rs, z_off0 + (uint32_t)((q >> 2) * 32 + (q & 3) * 8) * ldz2, 0, 0);
      return make_uint4(r.x, r.y, r.z, r.w);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) { rb[s][0] = ld_b(s, 0); rb[s][1] = ld_b(s, 1); }

    // ---- main loop: KS steps of 16 reduce elements, 8 MFMAs each ----
    uint4 fa[2][4];
    auto rd_a = [&](int kk, int buf) __attribute__((always_inline)) {
      const uint32_t a = a_base ^ (uint32_t)(kk << 5);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const uint4*>(smem + a + i * 32 * ROWB);
    };
    (void)rd_a;
    // (opaque per chunk: otherwise every step's fragment address - loop invariant - is hoisted out of the chunk loop, 64 live registers)
    uint32_t a_cur = a_base;
    asm volatile("" : "+v"(a_cur));
    {
      const uint32_t a = a_cur;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = *reinterpret_cast<const uint4*>(smem + a + i * 32 * ROWB);
    }
    __builtin_amdgcn_sched_barrier(0);
    // The instruction order IS the source order (sched_barrier(0) after every MFMA; left to itself the scheduler hoists the fragment
    // reads and ring refills of many steps to the top and spills 90-150 registers): behind MFMA m of a step the next step's
    // activation fragment m (m < 4), behind MFMAs 4 and 7 the two refills of the ring slot the step is consuming (weight-fragment major order: fragment 0 is free after MFMA 3).
    for_seq<0, KS>([&](auto ktag) __attribute__((always_inline)) {
      constexpr int kk = decltype(ktag)::value, cur = kk & 1, slot = kk % PF;
      const uint32_t an = a_cur ^ (uint32_t)((kk + 1) << 5);
      for_seq<0, 8>([&](auto mtag) __attribute__((always_inline)) {
        constexpr int mm = decltype(mtag)::value, j = mm >> 2, i = mm & 3;   // (weight fragment 0 is free behind MFMA 3)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot][j]), __builtin_bit_cast(bf16x8, fa[cur][i]), acc[i][j], 0, 0, 0);
        if constexpr (kk + 1 < KS && mm < 4) fa[cur ^ 1][mm] = *reinterpret_cast<const uint4*>(smem + an + mm * 32 * ROWB);
        if constexpr (mm == 4 || mm == 7) {
          constexpr int jj = mm == 4 ? 0 : 1;
          if constexpr (kk + PF < KS) rb[slot][jj] = ld_b(kk + PF, jj);
          else if constexpr (MODE == 1) rb[slot][jj] = ld_z(2 * slot + jj);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    // ---- epilogue, 32 rows at a time through the wave's own scratch ----
    for_seq<0, 4>([&](auto itag) __attribute__((always_inline)) {
      constexpr int i = decltype(itag)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack_bf16x2(acc[i][j][g * 4], acc[i][j][g * 4 + 1]);
          w.y = pack_bf16x2(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          *reinterpret_cast<uint2*>(scr + l31 * 128 + (((j * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = w;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for_seq<0, 4>([&](auto ptag) __attribute__((always_inline)) {
        constexpr int pp = decltype(ptag)::value;
        constexpr int rstep = i * 32 + pp * 8;               // rows below this lane's first row (n0 + (lane >> 3))
        const uint4 zz = *reinterpret_cast<const uint4*>(scr + (pp * 8 + (lane >> 3)) * 128 + (((lane & 7) ^ (lane >> 3)) << 4));
        float v[8];
        { const uint32_t w_[4] = {zz.x, zz.y, zz.z, zz.w}; unpack_words<bf16_t, 8>(w_, v); }
        // stores: buffer offsets = the chunk's lane offset + a wave-uniform row step; rows >= N fall outside the resource
        const uint32_t coff = c_off0 + (uint32_t)rstep * ldc2;
        const uint64_t didx = d_idx0 + (uint64_t)((long)rstep * p.M);
        if constexpr (MODE == 0) {
          if (p.Z) {
            const pg_u32x4 zu = {zz.x, zz.y, zz.z, zz.w};
            __builtin_amdgcn_raw_buffer_store_b128(zu, rz_rs, z_off0 + (uint32_t)rstep * ldz2, 0, 2);   // (nt: not read again before the backward pass)
          }
          act_fwd_n<ACT, 8>(v);
        } else {
          constexpr int q = i * 4 + pp;
          float zf[8];
          { const uint32_t w_[4] = {rb[q >> 1][q & 1].x, rb[q >> 1][q & 1].y, rb[q >> 1][q & 1].z, rb[q >> 1][q & 1].w}; unpack_words<bf16_t, 8>(w_, zf); }
          act_grad_mul_n<ACT, 8>(v, zf);
        }
        if (p.dthresh) dropout_apply<8>(v, dseed, didx, p.dthresh, p.dscale);
        const pg_u32x4 cu = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (p.nt & 2) __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 0);
      });
    });
  }
}

// host side: one launcher per mode (gemm_panel.hip: forward, gemm_panel_bwd.hip: act-grad), K and the activation by switch
int launch_panel_fwd(const PanelParams& p, int K, int act, hipStream_t s);
int launch_panel_actgrad(const PanelParams& p, int K, int act, hipStream_t s);

template <int MODE>
static int launch_panel_mode(const PanelParams& p, int K, int act, hipStream_t s) {
  const dim3 grid((p.N + 127) / 128), block(512);
#define SMX_PANEL_CASE(KK, AA) \
  if (K == KK && act == AA) { hipLaunchKernelGGL((gemm_panel_kernel<KK, MODE, AA>), grid, block, 0, s, p); return check_launch("smx_gemm_panel"); }
  SMX_PANEL_CASE(256, SMX_ACT_NONE) SMX_PANEL_CASE(256, SMX_ACT_SWISH) SMX_PANEL_CASE(256, SMX_ACT_GELU) SMX_PANEL_CASE(256, SMX_ACT_RELU)
  SMX_PANEL_CASE(512, SMX_ACT_NONE) SMX_PANEL_CASE(512, SMX_ACT_SWISH) SMX_PANEL_CASE(512, SMX_ACT_GELU) SMX_PANEL_CASE(512, SMX_ACT_RELU)
#undef SMX_PANEL_CASE
  return fail(SMX_EUNSUPPORTED, "smx_gemm_panel: K = %d / activation %d has no instantiation", K, act);
}

}  // namespace smx
