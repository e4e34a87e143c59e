// gemm_panel_w32.h — EXPERIMENT: the panel-resident GEMM with THREE waves per SIMD (12 waves = 768 threads, 168 registers each), every
// wave owning 128 rows x 32 columns at a time (64 accumulator registers, ONE weight fragment per step).  Same packed image, same
// arithmetic as gemm_panel.h; include it behind gemm_panel.h with -DSMX_PANEL_W32 (tools/experiments/panel_variant.sh).
#pragma once
// (included behind gemm_panel.h: see the header comment)

namespace smx {

template <int K, int MODE, int ACT>
__global__ __launch_bounds__(768) void gemm_panel_w32_kernel(PanelParams p) {
  constexpr int KS = K / 16, ROWB = K * 2, A_BYTES = 128 * ROWB, SCR = 2048, PF = 8, NW = 12;
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + NW * SCR];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = (int)(blockIdx.x / (unsigned)p.csplit) * 128, csi = (int)(blockIdx.x % (unsigned)p.csplit);
  const __amdgpu_buffer_rsrc_t rb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Bp), (short)0, (int)((long)p.M * (K + 16) * 2), 0x00020000);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  uint4 rb[PF], bfrag;
  auto ld_b = [&](uint4& dst, uint32_t voff, int kk) __attribute__((always_inline)) { panel_ld(dst, voff, rb_rs, (uint32_t)kk * 1024u); };
  {
    constexpr int CPR = K / 8, TOT = 128 * CPR, NA = (TOT + 767) / 768;
    const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0,
                                                                            (int)((((long)p.N - 1) * p.lda + K) * 2), 0x00020000);
    uint4 ra[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = min(t + 768 * i, TOT - 1), row = v / CPR, c = v % CPR, n = min(n0 + row, p.N - 1);
      const pg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra_rs, (uint32_t)(((long)n * p.lda + c * 8) * 2), 0, 0);
      const bool keep = p.row_mask ? p.row_mask[n] != 0 : true;
      ra[i] = keep ? make_uint4(r.x, r.y, r.z, r.w) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + 768 * i, row = v / CPR, c = v % CPR;
      if (v < TOT) *reinterpret_cast<uint4*>(smem + row * ROWB + ((c ^ (row & 15)) << 4)) = ra[i];
    }
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)((((long)p.N - 1) * p.ldc + p.M) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz_rs = __builtin_amdgcn_make_buffer_rsrc(p.Z ? p.Z : p.C, (short)0, (int)((((long)p.N - 1) * (p.Z ? p.ldz : p.ldc) + p.M) * 2), 0x00020000);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u, ldz2 = (uint32_t)p.ldz * 2u;
  const uint32_t a_base = (uint32_t)(l31 * ROWB + ((hi ^ (l31 & 15)) << 4));
  char* scr = smem + A_BYTES + wave * SCR;
  const uint64_t dseed = p.dthresh ? epoch_seed(p.seed, p.epoch) : 0;
  const uint32_t hm0 = mix32((uint32_t)dseed) ^ (uint32_t)(dseed >> 32), t16 = p.dthresh >> 16;
  const float dsc_d = (p.dthresh ? p.dscale : 1.f) * p.alpha, dsc_n = p.alpha;
  const int nch = p.M >> 5;                         // 32-column chunks = column blocks of the packed image

#pragma unroll 1
  for (int ch = wave + NW * csi; ch < nch; ch += NW * p.csplit) {
    uint32_t b_cur = lane16 + (uint32_t)ch * (uint32_t)((KS + 1) * 1024);
    asm volatile("" : "+v"(b_cur));
#pragma unroll
    for (int s = 0; s < PF; ++s) ld_b(rb[s], b_cur, s);
    if constexpr (MODE == 0) ld_b(bfrag, b_cur, KS);
    f32x16 acc[4];
    if constexpr (MODE == 0) {
      const uint8_t* mkp = p.row_mask;
      asm volatile("" : "+s"(mkp));
      uint32_t zr = 0u;
      asm volatile("" : "+v"(zr));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool keep = !hi && (mkp ? mkp[min(n0 + i * 32 + l31, p.N - 1)] != 0 : true);
        const uint4 ones = make_uint4(keep ? 0x3f803f80u : 0u, zr, zr, zr);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfrag), __builtin_bit_cast(bf16x8, ones), zero, 0, 0, 0);
      }
    }
    // readback layout of an item: 16 rows x 32 columns; lane -> row (lane >> 2), 8 columns (lane & 3) * 8
    uint32_t z_off0 = (uint32_t)(((long)(n0 + (lane >> 2)) * p.ldz + ch * 32 + (lane & 3) * 8) * 2);
    asm volatile("" : "+v"(z_off0));
    auto ld_z = [&](uint4& dst, int q) __attribute__((always_inline)) {   // item q (0..7): rows q * 16 + (lane >> 2)
      panel_ld(dst, z_off0 + (uint32_t)(q * 16) * ldz2, rz_rs, 0u);
    };
    uint4 fa[4];
    uint32_t a_cur = a_base;
    asm volatile("" : "+v"(a_cur));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + a_cur + i * 32 * ROWB);
    __builtin_amdgcn_sched_barrier(0);
    for_seq<0, KS>([&](auto ktag) __attribute__((always_inline)) {
      constexpr int kk = decltype(ktag)::value, slot = kk % PF;
      const uint32_t an = a_cur ^ (uint32_t)((kk + 1) << 5);
      for_seq<0, 4>([&](auto mtag) __attribute__((always_inline)) {
        constexpr int i = decltype(mtag)::value;
        if constexpr (MODE == 1 && kk == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot]), __builtin_bit_cast(bf16x8, fa[i]), zero, 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[slot]), __builtin_bit_cast(bf16x8, fa[i]), acc[i], 0, 0, 0);
        }
        if constexpr (kk + 1 < KS) fa[i] = *reinterpret_cast<const uint4*>(smem + an + i * 32 * ROWB);
        if constexpr (i == 3) {
          if constexpr (kk + PF < KS) ld_b(rb[slot], b_cur, kk + PF);
          else if constexpr (MODE == 1) ld_z(rb[slot], slot);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    uint32_t c_off0 = (uint32_t)(((long)(n0 + (lane >> 2)) * p.ldc + ch * 32 + (lane & 3) * 8) * 2);
    uint32_t d_p0 = ((uint32_t)(n0 + (lane >> 2)) * (uint32_t)p.drop_cols + (uint32_t)(ch * 32 + (lane & 3) * 8)) >> 1;
    const bool dchunk = p.dthresh != 0 && ch * 32 < p.drop_cols;
    const float dsc = dchunk ? dsc_d : dsc_n, dinv = 1.f / dsc;
    // scratch: 32 rows of 64 bytes; 16-byte chunk u of row r at position u ^ ((r >> 1) & 3)
    uint32_t s_wr = (uint32_t)(l31 * 64 + hi * 8), s_x = (uint32_t)((l31 >> 1) & 3);
    uint32_t s_rd = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4));
    asm volatile("" : "+v"(c_off0), "+v"(d_p0), "+v"(s_wr), "+v"(s_x), "+v"(s_rd));
    for_seq<0, 4>([&](auto itag) __attribute__((always_inline)) {
      constexpr int i = decltype(itag)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack_bf16x2(acc[i][g * 4], acc[i][g * 4 + 1]);
        w.y = pack_bf16x2(acc[i][g * 4 + 2], acc[i][g * 4 + 3]);
        *reinterpret_cast<uint2*>(scr + s_wr + (((uint32_t)g ^ s_x) << 4)) = w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for_seq<0, 2>([&](auto ptag) __attribute__((always_inline)) {
        constexpr int pp = decltype(ptag)::value, rstep = i * 32 + pp * 16, q = i * 2 + pp;
        const uint4 zz = *reinterpret_cast<const uint4*>(scr + s_rd + pp * 16 * 64);
        float v[8];
        { const uint32_t w_[4] = {zz.x, zz.y, zz.z, zz.w}; unpack_words<bf16_t, 8>(w_, v); }
        const uint32_t coff = c_off0 + (uint32_t)rstep * ldc2;
        if constexpr (MODE == 0) {
          if (p.Z) {
            const pg_u32x4 zu = {zz.x, zz.y, zz.z, zz.w};
            __builtin_amdgcn_raw_buffer_store_b128(zu, rz_rs, z_off0 + (uint32_t)rstep * ldz2, 0, 2);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = panel_act_scaled<ACT>(v[e], dsc, dinv);
        } else {
          float zf[8];
          { const uint32_t w_[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w}; unpack_words<bf16_t, 8>(w_, zf); }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= act_grad_c<ACT>(zf[e]) * dsc;
        }
        if (dchunk) panel_dropout8(v, hm0, d_p0 + (uint32_t)(rstep / 2) * (uint32_t)p.drop_cols, t16);
        const pg_u32x4 cu = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (p.nt & 2) __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(cu, rc_rs, coff, 0, 0);
      });
    });
  }
}

}  // namespace smx
