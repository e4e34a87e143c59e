// gemm_panel2.h — EXPERIMENT (tools/experiments/panel_variant.sh v2 "-DSMX_PANEL_V2"): the panel-resident GEMM with ONE wave per SIMD
// (4 waves, 512 registers each) that interleaves the epilogue of its previous column chunk, piece by piece, between the MFMAs of
// the current chunk's main loop - instead of relying on a second wave per SIMD to run beside it (gemm_panel.h: the two waves of a
// SIMD overlap poorly, ~33 K cycles per pair of chunks against 16 K of MFMA issue and 18 K of VALU issue).
// At the end of a chunk's main loop the accumulators are converted to packed bf16 (64 registers, `pk`); during the NEXT chunk's
// main loop ~100 small pieces (LDS transposition writes, a read, 4-element activation groups, dropout pairs, pack + store) are
// issued one behind (nearly) every MFMA.  The first chunk runs the pieces on garbage with out-of-range store offsets (branch-free:
// a conditional store would make hipcc's vmcnt merge conservative), a drain pass finishes the last chunk.
#pragma once
#include "gemm_panel.h"

namespace smx {

template <int K, int MODE, int ACT>
__global__ __launch_bounds__(256) void gemm_panel2_kernel(PanelParams p) {
  constexpr int KS = K / 16, ROWB = K * 2, A_BYTES = 128 * ROWB, SCR = 4096, PF = 8, NW = 4;
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + NW * SCR];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = (int)(blockIdx.x / (unsigned)p.csplit) * 128, csi = (int)(blockIdx.x % (unsigned)p.csplit);

  const __amdgpu_buffer_rsrc_t rb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Bp), (short)0, (int)((long)p.M * (K + 16) * 2), 0x00020000);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  uint4 rb[PF][2], bfrag[2];
  auto ld_b = [&](uint4& dst, uint32_t voff, int kk, int j) __attribute__((always_inline)) {
    panel_ld(dst, voff, rb_rs, (uint32_t)(j * (KS + 1) + kk) * 1024u);
  };
  auto chunk_voff = [&](int c) __attribute__((always_inline)) -> uint32_t { return lane16 + (uint32_t)c * (uint32_t)(2 * (KS + 1) * 1024); };

  {   // the panel (two halves: 64 registers of staging at a time)
    constexpr int CPR = K / 8, NA = 128 * CPR / 256;
    const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0,
                                                                            (int)((((long)p.N - 1) * p.lda + K) * 2), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 ra[NA / 2];
#pragma unroll
      for (int i = 0; i < NA / 2; ++i) {
        const int v = t + 256 * (i + h * (NA / 2)), row = v / CPR, c = v % CPR, n = min(n0 + row, p.N - 1);
        const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra_rs, (uint32_t)(((long)n * p.lda + c * 8) * 2), 0, 0);
        const bool keep = p.row_mask ? p.row_mask[n] != 0 : true;
        ra[i] = keep ? make_uint4(r.x, r.y, r.z, r.w) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int i = 0; i < NA / 2; ++i) {
        const int v = t + 256 * (i + h * (NA / 2)), row = v / CPR, c = v % CPR;
        *reinterpret_cast<uint4*>(smem + row * ROWB + ((c ^ (row & 15)) << 4)) = ra[i];
      }
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)((((long)p.N - 1) * p.ldc + p.M) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz_rs = __builtin_amdgcn_make_buffer_rsrc(p.Z ? p.Z : p.C, (short)0, (int)((((long)p.N - 1) * (p.Z ? p.ldz : p.ldc) + p.M) * 2), 0x00020000);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u, ldz2 = (uint32_t)(p.Z ? p.ldz : p.ldc) * 2u;
  const uint32_t a_base = (uint32_t)(l31 * ROWB + ((hi ^ (l31 & 15)) << 4));
  char* scr = smem + A_BYTES + wave * SCR;
  const uint64_t dseed = p.dthresh ? epoch_seed(p.seed, p.epoch) : 0;
  const uint32_t hm0 = mix32((uint32_t)dseed) ^ (uint32_t)(dseed >> 32), t16 = p.dthresh >> 16;
  const float dsc_d = (p.dthresh ? p.dscale : 1.f) * p.alpha, dsc_n = p.alpha;
  const int nch = p.M >> 6;
  const uint32_t s_wr = (uint32_t)(l31 * 128 + hi * 8), s_x = (uint32_t)(l31 & 7), s_rd = (uint32_t)((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4));

  // ---- state of the PREVIOUS chunk (whose epilogue runs inside the current main loop) ----
  uint32_t pk[4][2][8];                          // its outputs, packed bf16 pairs in accumulator layout
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) pk[i][j][e] = 0u;
  uint4 zr[MODE == 1 ? 16 : 1];                  // its saved pre-activation items (MODE 1)
  uint32_t pc_off = 0x80000000u, pz_off = 0x80000000u, pd_p0 = 0u;   // first chunk: every store out of range (dropped)
  bool pdchunk = false;
  float pdsc = 1.f, pdinv = 1.f;
  // epilogue pieces of one chunk: per 32-row block i: piece 0 = the 8 transposition writes, then per item pp (8 rows x 64 columns of
  // this wave: one 16-byte row segment per lane) 6 pieces: read | unpack (+ Z store / z unpack) + act 0-3 | act 4-7 | dropout pairs 0-1 |
  // pairs 2-3 | pack + store
  constexpr int NPIECE = 4 * 25;
  uint4 e_zz; float e_v[8]; float e_zf[MODE == 1 ? 8 : 1];
  auto piece = [&](auto utag) __attribute__((always_inline)) {
    constexpr int U = decltype(utag)::value, i = U / 25, r = U % 25;
    if constexpr (r == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(scr + s_wr + (((uint32_t)(j * 4 + g) ^ s_x) << 4)) = make_uint2(pk[i][j][2 * g], pk[i][j][2 * g + 1]);
    } else {
      constexpr int pp = (r - 1) / 6, sub = (r - 1) % 6, rstep = i * 32 + pp * 8, q = i * 4 + pp;
      if constexpr (sub == 0) {
        e_zz = *reinterpret_cast<const uint4*>(scr + s_rd + pp * 8 * 128);
      } else if constexpr (sub == 1) {
        { const uint32_t w_[4] = {e_zz.x, e_zz.y, e_zz.z, e_zz.w}; unpack_words<bf16_t, 8>(w_, e_v); }
        if constexpr (MODE == 0) {
          const pg_u32x4 zu = {e_zz.x, e_zz.y, e_zz.z, e_zz.w};
          __builtin_amdgcn_raw_buffer_store_b128(zu, rz_rs, pz_off + (uint32_t)rstep * ldz2, 0, 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) e_v[e] = panel_act_scaled<ACT>(e_v[e], pdsc, pdinv);
        } else {
          { const uint32_t w_[4] = {zr[q].x, zr[q].y, zr[q].z, zr[q].w}; unpack_words<bf16_t, 8>(w_, e_zf); }
#pragma unroll
          for (int e = 0; e < 4; ++e) e_v[e] *= act_grad_c<ACT>(e_zf[e]) * pdsc;
        }
      } else if constexpr (sub == 2) {
        if constexpr (MODE == 0) {
#pragma unroll
          for (int e = 4; e < 8; ++e) e_v[e] = panel_act_scaled<ACT>(e_v[e], pdsc, pdinv);
        } else {
#pragma unroll
          for (int e = 4; e < 8; ++e) e_v[e] *= act_grad_c<ACT>(e_zf[e]) * pdsc;
        }
      } else if constexpr (sub == 3 || sub == 4) {
        if (pdchunk) {
          const uint32_t p0 = pd_p0 + (uint32_t)(rstep / 2) * (uint32_t)p.drop_cols;
          const uint32_t ph = p0 ^ hm0 ^ pair_hi_mix(p0);
#pragma unroll
          for (int q2 = (sub - 3) * 2; q2 < (sub - 3) * 2 + 2; ++q2) {
            const uint32_t h = mix32_1(ph ^ (uint32_t)q2);
            e_v[2 * q2] = (h & 0xffffu) >= t16 ? e_v[2 * q2] : 0.f;
            e_v[2 * q2 + 1] = (h >> 16) >= t16 ? e_v[2 * q2 + 1] : 0.f;
          }
        }
      } else {
        const pg_u32x4 cu = {pack_bf16x2(e_v[0], e_v[1]), pack_bf16x2(e_v[2], e_v[3]), pack_bf16x2(e_v[4], e_v[5]), pack_bf16x2(e_v[6], e_v[7])};
        const uint32_t coff = pc_off + (uint32_t)rstep * ldc2;
        if (p.nt & 2) __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 0);
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll 1
  for (int ch = wave + NW * csi; ch < nch; ch += NW * p.csplit) {
    uint32_t b_cur = chunk_voff(ch);
    asm volatile("" : "+v"(b_cur));
#pragma unroll
    for (int s = 0; s < PF; ++s) { ld_b(rb[s][0], b_cur, s, 0); ld_b(rb[s][1], b_cur, s, 1); }
    if constexpr (MODE == 0) { ld_b(bfrag[0], b_cur, KS, 0); ld_b(bfrag[1], b_cur, KS, 1); }
    if constexpr (MODE == 0) {
      const uint8_t* mkp = p.row_mask;
      asm volatile("" : "+s"(mkp));
      uint32_t zr0 = 0u;
      asm volatile("" : "+v"(zr0));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool keep = !hi && (mkp ? mkp[min(n0 + i * 32 + l31, p.N - 1)] != 0 : true);
        const uint4 ones = make_uint4(keep ? 0x3f803f80u : 0u, zr0, zr0, zr0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfrag[j]), __builtin_bit_cast(bf16x8, ones), zero, 0, 0, 0);
      }
    }
    uint32_t z_off0 = (uint32_t)(((long)(n0 + (lane >> 3)) * (p.Z ? p.ldz : p.ldc) + ch * 64 + (lane & 7) * 8) * 2);
    asm volatile("" : "+v"(z_off0));
    uint4 fa[4];
    uint32_t a_cur = a_base;
    asm volatile("" : "+v"(a_cur));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + a_cur + i * 32 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
    for_seq<0, KS>([&](auto ktag) __attribute__((always_inline)) {
      constexpr int kk = decltype(ktag)::value, slot = kk % PF;
      const uint32_t an = a_cur ^ (uint32_t)((kk + 1) << 5);
      for_seq<0, 8>([&](auto mtag) __attribute__((always_inline)) {
        constexpr int mm = decltype(mtag)::value, i = mm >> 1, j = mm & 1, S = kk * 8 + mm, NSLOT = KS * 8;
        if constexpr (MODE == 1 && kk == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot][j]), __builtin_bit_cast(bf16x8, fa[i]), zero, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot][j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
        }
        if constexpr (kk + 1 < KS && j == 1) fa[i] = *reinterpret_cast<const uint4*>(smem + an + i * 32 * ROWB);
        if constexpr (mm >= 6) {
          constexpr int jj = mm - 6;
          if constexpr (kk + PF < KS) ld_b(rb[slot][jj], b_cur, kk + PF, jj);
        }
        // the previous chunk's epilogue pieces u with floor(u * NSLOT / NPIECE) == S behind this MFMA
        for_seq<0, NPIECE>([&](auto utag) __attribute__((always_inline)) {
          constexpr int U = decltype(utag)::value;
          if constexpr (U * NSLOT / NPIECE == S) {
            piece(utag);
            if constexpr (MODE == 1 && U % 25 >= 1 && (U % 25 - 1) % 6 == 2) {     // the item's saved pre-activation was consumed (sub-piece 1): request THIS chunk's
              constexpr int q = (U / 25) * 4 + (U % 25 - 1) / 6;
              panel_ld(zr[q], z_off0 + (uint32_t)((q >> 2) * 32 + (q & 3) * 8) * ldz2, rz_rs, 0u);
            }
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // ---- this chunk becomes the previous one: accumulators -> packed bf16, its offsets ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[i][j][e] = pack_bf16x2(acc[i][j][2 * e], acc[i][j][2 * e + 1]);
    pc_off = (uint32_t)(((long)(n0 + (lane >> 3)) * p.ldc + ch * 64 + (lane & 7) * 8) * 2);
    pz_off = (MODE == 0 && !p.Z) ? 0x80000000u : z_off0;
    pd_p0 = ((uint32_t)(n0 + (lane >> 3)) * (uint32_t)p.drop_cols + (uint32_t)(ch * 64 + (lane & 7) * 8)) >> 1;
    pdchunk = p.dthresh != 0 && ch * 64 < p.drop_cols;
    pdsc = pdchunk ? dsc_d : dsc_n;
    pdinv = 1.f / pdsc;
  }
  // ---- drain: the last chunk's epilogue ----
  for_seq<0, NPIECE>([&](auto utag) __attribute__((always_inline)) { piece(utag); });
}

}  // namespace smx
