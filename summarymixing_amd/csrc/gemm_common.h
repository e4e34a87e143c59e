// gemm_common.h — pieces shared by the GEMM kernels of libsmx.so (gemm.hip: tiled register-staged kernels and the
// wgrad LDS-DMA kernel; pgemm.hip: the persistent LDS-DMA kernel): parameter block, LDS images and fragment reads,
// the staged epilogue.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "smx_common.h"

namespace smx {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct GemmParams {
  const void* A; const void* B; void* C;
  long lda, ldb, ldc, sA, sB, sC;
  int N, M, K, batch, splits, kchunk;
  int tiles_n, tiles_m;
  smx_epilogue e;
  int epi_vec;
  int epi_lds;      // outputs are 16-byte addressable: stage the tile through LDS and store whole rows
  long sSplit;      // element stride between split-K slabs of C (out_mode F32)
  float* acolsum;                   // TN only: [splits][batch][N] partial column sums of the A operand (bias gradient)
  const uint64_t* epoch;            // device step counter mixed into the dropout seed (or null)
  unsigned dthresh; float dscale;   // fused dropout: drop if hash16 < dthresh>>16, survivors * dscale
  int drop_cols;                    // ... on output columns m < drop_cols, mask index n * drop_cols + m (= M: all columns)
  long long* dbg;   // debug: per-wave s_memtime stamps (smx_debug_set_timing_buffer)
  int nt;       // non-temporal store hints: 1 = saved pre-activation Z, 2 = the output C
  int epi_simple;  // no element-wise side input, no column sums: the SIMPLE instantiation of epilogue_phase
  int ablate;   // debug (env SMX_GEMM_ABLATE): 1 = no epilogue stores, 2 = no MFMA, 4 = no global loads
  smx_gemm_plan* plan;   // smx_gemm_plan_query: the dispatch records the instantiation here instead of launching it
  // implicit 3x3 / stride 2 / reflect-pad-1 patch matrix (gemm_kernel<..., GATHER>): the operand "rows x 9 C" is never
  // materialised - row n = (b, t2, f2), columns [tap * 64, tap * 64 + 64) = the 64 channels of input pixel
  // (reflect(2 t2 + dt - 1), reflect(2 f2 + df - 1)) of the channels-last tensor (B, g_T, g_F, 64)
  int g_T, g_F, g_T2, g_F2, gather;   // gather: 0 none, 1 A operand (NT), 2 B operand (TN)
  long g_npix;                        // B * g_T * g_F
};

// every launch site of the GEMM dispatch: `if (plan_only(p, ...)) return SMX_OK;` in front of its hipLaunchKernelGGL
// rows per tile of the LayerNorm-fused GEMMs of width M for N frames (= rows per dgamma / dbeta partial row pair).  d_model = 256 has a
// 64-row tile (round 6): two workgroups per CU = rounds of 512, and a mostly empty last round costs a whole one, so the choice is a
// quantisation problem: cost = rounds x rows, a workgroup that has its CU to itself (<= 256 of them) runs ~1/4 faster.  Same-box
// A/B of the C2b step (128 | 64 rows, ms): B = 36: 9.42 | 8.90, 48: 10.08 | 9.51, 64: 11.28 | 10.76, 72: 14.58 | 14.68, 80: 14.65 | 14.94,
// 96: 15.49 | 16.30, 128: 17.72 | 18.58.  SMX_LN_TILE64: 0 never, 1 this rule, 2 always.
#ifndef SMX_R64_OCC
#define SMX_R64_OCC 2
#endif
inline int ln_tile_rows_for(int N, int M) {
  const int mode = cfg().ln_tile64;
  if (M != 256 || mode == 0 || N <= 0) return 128;
  if (mode == 2) return 64;
  auto cost = [&](int rows) {
    const long w = (N + rows - 1) / rows;
    const long slots = rows == 64 ? 256L * SMX_R64_OCC : 512L;
    return (double)((w + slots - 1) / slots) * rows * (w <= 256 ? 0.75 : 1.0);
  };
  return cost(64) < cost(128) - 1e-9 ? 64 : 128;
}
inline bool plan_only(GemmParams& p, int kernel, bool a_kc, bool b_kc, int tn, int tm, bool vec, int lnf, int gather) {
  if (!p.plan) return false;
  *p.plan = smx_gemm_plan{kernel, a_kc, b_kc, tn, tm, vec, lnf, gather};
  return true;
}

// the LayerNorm-fused row-complete tiles (gemm_ln256.hip / gemm_ln512.hip)
int launch_ln_fused_256(GemmParams& p, bool b_kc, hipStream_t s);
int launch_ln_fused_512(GemmParams& p, bool b_kc, hipStream_t s);

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> { static constexpr int BK = 64; static constexpr int VPT = 8; };
template <> struct ElemTraits<float>  { static constexpr int BK = 32; static constexpr int VPT = 4; };

// ---- guarded 16-byte fetch of VPT consecutive elements -------------------------------------------------
template <typename T, bool VEC>
__device__ __forceinline__ uint4 ld_contig(const T* p, int nvalid, const T* safe) {
  constexpr int VPT = ElemTraits<T>::VPT;
  if constexpr (VEC) {
    // vector mode: the host guarantees whole, aligned vectors, so a lane is either fully in or fully out.
    // Branch-free: out-of-range lanes read a valid dummy address and are zeroed by a select, so that all the
    // loads of a tile are issued back to back (an exec-masked branch would cost one s_waitcnt vmcnt(0) each).
    const bool ok = nvalid >= VPT;
    uint4 r = *reinterpret_cast<const uint4*>(ok ? p : safe);
    return ok ? r : make_uint4(0, 0, 0, 0);
  } else {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (nvalid <= 0) return r;
    uint32_t w[4] = {0, 0, 0, 0};
    if constexpr (sizeof(T) == 2) {
      const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < nvalid) w[i >> 1] |= (uint32_t)q[i] << ((i & 1) * 16);
    } else {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nvalid) w[i] = q[i];
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---- stage one operand tile (ROWS x BK) : global -> registers ------------------------------------------
// KC: element (row r, reduce k) at base[r*ld + k];  KS: at base[k*ld + r]
template <typename T, bool KC, int ROWS, bool VEC>
__device__ __forceinline__ void stage_load(uint4 (&reg)[ROWS / 32], const T* base, long ld, int row0, int rows_total,
                                           int k0, int k1, int t) {
  constexpr int BK = ElemTraits<T>::BK;
  constexpr int VPT = ElemTraits<T>::VPT;
  if constexpr (sizeof(T) == 2 && KC) {
    constexpr int NV = ROWS / 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, c = v & 7;
      int rg = row0 + row, kg = k0 + c * 8;
      int nv = (rg < rows_total) ? (k1 - kg) : 0;
      reg[i] = ld_contig<T, VEC>(base + (long)rg * ld + kg, nv, base);
    }
  } else if constexpr (sizeof(T) == 2 && !KC) {
    constexpr int RC = ROWS / 8;           // 16-byte row chunks per k row
    constexpr int NV = ROWS / 32;          // vectors per thread (64 k rows x RC chunks / 256 threads)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = t + 256 * i, rc = v % RC, k = v / RC;
      const int rg = row0 + rc * 8, kg = k0 + k;
      reg[i] = ld_contig<T, VEC>(base + (long)kg * ld + rg, kg < k1 ? rows_total - rg : 0, base);
    }
  } else if constexpr (sizeof(T) == 4 && KC) {
    constexpr int NV = ROWS / 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, k4 = v & 7;
      int rg = row0 + row, kg = k0 + k4 * 4;
      int nv = (rg < rows_total) ? (k1 - kg) : 0;
      reg[i] = ld_contig<T, VEC>(base + (long)rg * ld + kg, nv, base);
    }
  } else {
    constexpr int NV = ROWS / 32;
    constexpr int R4 = ROWS / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, r4 = v % R4, k = v / R4;
      int rg = row0 + r4 * 4, kg = k0 + k;
      reg[i] = ld_contig<T, VEC>(base + (long)kg * ld + rg, kg < k1 ? rows_total - rg : 0, base);
    }
  }
  (void)BK; (void)VPT;
}

// ---- the same stage through buffer loads (bf16, aligned shapes) ---------------------------------------------------------
// PMC (profiles/r01_pmc_sq.txt): a bias-only K=256 tile executed ~1270 VALU instructions per wave next to its 64 MFMAs,
// about half of them in the main loop - per-load 64-bit address arithmetic and bounds predicates that the compiler
// re-materialises every K step at the 168-register budget.  With a buffer resource the per-thread byte offsets of a
// stage are computed ONCE (one 32-bit VGPR per vector), the K step advances through the scalar offset, and the bounds
// check is the hardware's (out-of-range rows / k rows return zeros): the main loop's loads need no VALU at all.
// Requirements (checked on the host, else the generic path): operand span < 2 GB; for a reduce-contiguous operand
// K % 64 == 0 (a k tail inside a row would read the next row instead of zeros).
template <typename T, bool KC, int ROWS>
struct BufStage {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[ROWS / 32];
  uint32_t kbytes;                                       // bytes per unit k
  __device__ __forceinline__ void init(const T* base, long ld, int row0, int rows_total, int K, int t) {
    const long span = KC ? ((long)(rows_total - 1) * ld + K) : ((long)(K - 1) * ld + rows_total);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), (short)0, (int)(span * (long)sizeof(T)), 0x00020000);
    kbytes = KC ? (uint32_t)sizeof(T) : (uint32_t)(ld * (long)sizeof(T));
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int v = t + 256 * i;
      if constexpr (KC) {
        const int rg = row0 + (v >> 3);
        voff[i] = rg < rows_total ? (uint32_t)(((long)rg * ld + (v & 7) * 8) * (long)sizeof(T)) : 0x80000000u;
      } else {
        constexpr int RC = ROWS / 8;
        const int rg = row0 + (v % RC) * 8;
        voff[i] = rg < rows_total ? (uint32_t)(((long)(v / RC) * ld + rg) * (long)sizeof(T)) : 0x80000000u;
      }
    }
  }
  __device__ __forceinline__ void load(uint4 (&reg)[ROWS / 32], int k0) const {
    const uint32_t soff = (uint32_t)k0 * kbytes;         // (uniform)
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], soff, 0);
      reg[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
  }
  // the same, issued whether the K tile exists or not (`valid` uniform): an invalid request carries bit 31 in its offset -
  // out of the buffer's range, zeros come back and nothing is fetched (branch-free main loops)
  __device__ __forceinline__ void load_pred(uint4 (&reg)[ROWS / 32], int k0, bool valid) const {
    const uint32_t soff = valid ? (uint32_t)k0 * kbytes : 0u;
    const uint32_t inval = valid ? 0u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i] | inval, soff, 0);
      reg[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
  }
};

// ---- implicit conv-patch operand (C = 64 channels = one tap per BK = 64 reduce elements / per 64-column tile) --------------
__device__ __forceinline__ int gather_reflect(int i, int L) { return i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i); }
// KC form (NT forward): the tile's ROWS patch rows, one 64-wide tap per K step.  Thread piece i: row v >> 3, 16-byte chunk v & 7.
template <int ROWS>
struct GatherStageKC {
  __amdgpu_buffer_rsrc_t rsrc;
  int tt[ROWS / 32], ff[ROWS / 32], bb[ROWS / 32];      // 2 t2 - 1, 2 f2 - 1, b * T * F (pixels); bb < 0: row beyond N
  uint32_t coff;
  __device__ __forceinline__ void init(const void* base, long npix, const GemmParams& p, int row0, int t) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)(npix * 128), 0x00020000);
    coff = (uint32_t)(t & 7) * 16;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int rg = row0 + ((t + 256 * i) >> 3);
      const int f2 = rg % p.g_F2, r2 = rg / p.g_F2, t2 = r2 % p.g_T2, b = r2 / p.g_T2;
      tt[i] = 2 * t2 - 1; ff[i] = 2 * f2 - 1;
      bb[i] = rg < p.N ? b * p.g_T * p.g_F : -1;
    }
  }
  __device__ __forceinline__ void load(uint4 (&reg)[ROWS / 32], int k0, const GemmParams& p) const {
    const int tap = k0 >> 6, dt = tap / 3, df = tap - dt * 3;  // (uniform)
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int pix = bb[i] + gather_reflect(tt[i] + dt, p.g_T) * p.g_F + gather_reflect(ff[i] + df, p.g_F);
      const uint32_t off = bb[i] >= 0 ? (uint32_t)pix * 128u + coff : 0x80000000u;
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      reg[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
  }
};
// reduce-strided form (TN weight gradient): the K rows of a stage are patch rows n = k0 + krow, the tile's 64 columns are ONE tap
template <int ROWS>
struct GatherStageKS {
  __amdgpu_buffer_rsrc_t rsrc;
  int dt, df, krow[ROWS / 32];
  uint32_t coff[ROWS / 32];
  __device__ __forceinline__ void init(const void* base, long npix, int m0, int t) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)(npix * 128), 0x00020000);
    const int tap = m0 >> 6;
    dt = tap / 3; df = tap - dt * 3;
    constexpr int RC = ROWS / 8;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int v = t + 256 * i;
      krow[i] = v / RC;
      coff[i] = (uint32_t)(v % RC) * 16;
    }
  }
  __device__ __forceinline__ void load(uint4 (&reg)[ROWS / 32], int k0, int kend, const GemmParams& p) const {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
      const int n = k0 + krow[i];
      const int f2 = n % p.g_F2, r2 = n / p.g_F2, t2 = r2 % p.g_T2, b = r2 / p.g_T2;
      const int pix = b * p.g_T * p.g_F + gather_reflect(2 * t2 - 1 + dt, p.g_T) * p.g_F + gather_reflect(2 * f2 - 1 + df, p.g_F);
      const uint32_t off = n < kend ? (uint32_t)pix * 128u + coff[i] : 0x80000000u;
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
      const u32v4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      reg[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
  }
};

// ---- folded DFT operand (float32 NT): the real DFT of a windowed frame x[0..n) with a symmetric window needs only
//   s[j] = x[j] + x[n - j] (cosine part, j = 0..n/2, the two end points alone) and d[j] = x[j] - x[n - j] (sine part, j = 1..n/2-1):
// two GEMMs of half the reduction length instead of one over n.  The A tile is built from TWO 16-byte loads of the frame (one
// of them walking backwards from x[n - j]) while it passes through the registers; frames are rows of the zero-padded waveform
// (leading dimension = hop).  MODE 3 = s (columns >= n/2 + 1 are zero), MODE 4 = d (column 0 is zero).
template <int MODE, int ROWS>
__device__ __forceinline__ void fold_stage_load(uint4 (&reg)[ROWS / 32], const float* base, long ld, int row0, int rows_total,
                                                int k0, int nfft, int t) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int half = nfft >> 1;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int v = t + 256 * i, row = v >> 3, k4 = v & 7;
    const int rg = row0 + row, kg = k0 + k4 * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (rg < rows_total && kg <= (MODE == 3 ? half : half - 1)) {
      const float* xr = base + (long)rg * ld;
      const f4u a = *reinterpret_cast<const f4u*>(xr + kg);
      const f4u b = *reinterpret_cast<const f4u*>(xr + nfft - kg - 3);       // x[n - j] for j = kg + 3 .. kg
      const float pa[4] = {a.x, a.y, a.z, a.w}, pb[4] = {b.w, b.z, b.y, b.x};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = kg + e;
        if (MODE == 3) o[e] = j > half ? 0.f : ((j == 0 || j == half) ? pa[e] : pa[e] + pb[e]);
        else o[e] = (j == 0 || j >= half) ? 0.f : pa[e] - pb[e];
      }
    }
    reg[i] = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
  }
}

// ---- registers -> LDS image -----------------------------------------------------------------------------
template <typename T, bool KC, int ROWS>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[ROWS / 32], char* lds, int t) {
  if constexpr (sizeof(T) == 2 && KC) {
    constexpr int NV = ROWS / 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, c = v & 7;
      *reinterpret_cast<uint4*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = reg[i];
    }
  } else if constexpr (sizeof(T) == 2 && !KC) {
    // plain copy into a [k][ROWS (+32 pad)] image: the transposition is done by ds_read_b64_tr_b16 on the way out
    constexpr int RC = ROWS / 8;
    constexpr int NV = ROWS / 32;
    constexpr int KSTRB = (ROWS + 32) * 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = t + 256 * i, rc = v % RC, k = v / RC;
      *reinterpret_cast<uint4*>(lds + k * KSTRB + rc * 16) = reg[i];
    }
  } else if constexpr (sizeof(T) == 4 && KC) {
    constexpr int NV = ROWS / 32;
    constexpr int KSTR = ROWS + 4;
    float* l = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, k4 = v & 7;
      const float* f = reinterpret_cast<const float*>(&reg[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) l[(k4 * 4 + j) * KSTR + row] = f[j];
    }
  } else {
    constexpr int NV = ROWS / 32;
    constexpr int R4 = ROWS / 4;
    constexpr int KSTR = ROWS + 4;
    float* l = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, r4 = v % R4, k = v / R4;
      *reinterpret_cast<uint4*>(l + k * KSTR + r4 * 4) = reg[i];
    }
  }
}

template <typename T, int ROWS, bool KC>
constexpr int lds_bytes() {
  return sizeof(T) == 2 ? (KC ? ROWS * 128 : 64 * (ROWS + 32) * 2) : 32 * (ROWS + 4) * 4;
}

// ---- fragment reads ---------------------------------------------------------------------------------------
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 frag_bf16(const char* lds, int r, int kk, int hi) {
  if constexpr (KC) {
    int c = kk * 2 + hi;
    uint4 v = *reinterpret_cast<const uint4*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    // reduce-strided operand: LDS holds [k][ROWS+32] (k rows of the tile, row index contiguous).  One
    // ds_read_b64_tr_b16 hands every lane of a 16-lane group the 4 consecutive k of ITS row (hardware 4x16
    // transpose; verified by tools/tr_probe.hip); two of them make the 8-k MFMA fragment.  Row stride +64 B keeps the
    // four k rows of a group and the neighbouring group on distinct banks.
    constexpr int KSTR = ROWS + 32;
    typedef short short4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) short4_t* lds_s4;
    const int lane = (r & 31) | (hi << 5);              // r carries the lane's row; rebuild the lane id
    const int li = lane & 15, g1 = (lane >> 4) & 1;
    const int rbase = r - (r & 31);                     // fragment row base
    const uint16_t* l16 = reinterpret_cast<const uint16_t*>(lds);
    const uint16_t* p0 = l16 + (kk * 16 + hi * 8 + (li >> 2)) * KSTR + rbase + g1 * 16 + (li & 3) * 4;
    const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
    const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * KSTR));
    const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
    uint4 v = make_uint4(ua.x, ua.y, ub.x, ub.y);
    return __builtin_bit_cast(bf16x8, v);
  }
}

// reduce-contiguous fragments with the loop-invariant part of the swizzled address precomputed: chunk kk*2 + hi of row r
// sits at r*128 + (((kk*2 + hi) ^ ((r >> 1) & 7)) << 4) = frag_pre(r, hi) ^ (kk << 5)  (r*128 has no bits below 128, and
// kk*2 only touches bits 1-2 of the chunk index) - one v_xor with an immediate per fragment and sub-step
__device__ __forceinline__ uint32_t frag_pre(int r, int hi) { return (uint32_t)(r * 128 + ((hi ^ ((r >> 1) & 7)) << 4)); }
__device__ __forceinline__ bf16x8 frag_kc(const char* lds, uint32_t pre, int kk) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (pre ^ (uint32_t)(kk << 5))));
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// activation of one accumulator quad: ONE uniform switch per 4 values (never per element)
__device__ __forceinline__ void act4(int act, float (&y)[4]) {
  switch (act) {
    case SMX_ACT_GELU:
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = act_fwd_c<SMX_ACT_GELU>(y[q]);
      break;
    case SMX_ACT_SWISH:
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = act_fwd_c<SMX_ACT_SWISH>(y[q]);
      break;
    case SMX_ACT_LEAKY_RELU:
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = act_fwd_c<SMX_ACT_LEAKY_RELU>(y[q]);
      break;
    case SMX_ACT_RELU:
#pragma unroll
      for (int q = 0; q < 4; ++q) y[q] = act_fwd_c<SMX_ACT_RELU>(y[q]);
      break;
    default: break;
  }
}

__device__ __forceinline__ long c0_row(const smx_epilogue& e, int n) {
  if (e.c0_mode == SMX_C0_GROUP) return n / e.c0_div;
  if (e.c0_mode == SMX_C0_MOD) return n % e.c0_div;
  return n;
}

// ---- one epilogue phase: WN staged fp32 rows (LDS) -> outputs.  vmcnt retires in order and counts stores too, so
// every side input of a batch of items (C0 / residual or saved pre-activation / mask) is requested BEFORE the
// batch's first store; the stores then stream out without any wave waiting on them.
// Every optional feature sits behind ONE wave-uniform branch per item (never per element), so an absent feature
// costs no VALU work: the epilogue is VALU-issue bound when the three resident workgroups of a CU reach it together.
// EVEC: whole, 16-byte aligned items (host-checked) -> no per-element guards at all. -------------------------------
template <int ACT, int CW>
__device__ __forceinline__ void act_fwd_n(float (&v)[CW]) {
#pragma unroll
  for (int q = 0; q < CW; ++q) v[q] = act_fwd_c<ACT>(v[q]);
}
template <int ACT, int CW>
__device__ __forceinline__ void act_grad_mul_n(float (&v)[CW], const float (&z)[CW]) {
#pragma unroll
  for (int q = 0; q < CW; ++q) v[q] *= act_grad_c<ACT>(z[q]);
}
// CW elements of type T held as raw 32-bit words -> floats
template <typename T, int CW>
__device__ __forceinline__ void unpack_words(const uint32_t (&w)[CW * sizeof(T) / 4], float (&f)[CW]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int q = 0; q < CW / 2; ++q) { f[2 * q] = bf16_bits_to_f32(w[q] & 0xffffu); f[2 * q + 1] = bf16_bits_to_f32(w[q] >> 16); }
  } else {
#pragma unroll
    for (int q = 0; q < CW; ++q) f[q] = __uint_as_float(w[q]);
  }
}
template <int NW>
__device__ __forceinline__ void ld_words(const void* p, uint32_t (&w)[NW]) {
  if constexpr (NW == 4) { const uint4 u = *reinterpret_cast<const uint4*>(p); w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w; }
  else { const uint2 u = *reinterpret_cast<const uint2*>(p); w[0] = u.x; w[1] = u.y; }
}
template <typename T, int CW>
__device__ __forceinline__ void st_elems(void* p, const float (&v)[CW]) {   // CW elements of type T, one store
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (CW == 8) {
    uint4 u;
    u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
  } else {
    uint2 u;
    u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
  }
}

// "consume" a loaded register: the compiler places the load's s_waitcnt HERE (zero instructions otherwise)
__device__ __forceinline__ void settle(uint32_t& w) { asm volatile("" : "+v"(w)); }
__device__ __forceinline__ void settle(float& w) { asm volatile("" : "+v"(w)); }

// side = [TILE_M] bias (0 when absent) followed by [TILE_N] row factors (row_mask * alpha), staged in LDS by the
// kernel prologue: reading them costs LDS (lgkmcnt) traffic only, never a vmcnt wait behind in-flight stores.
// same, with the non-temporal hint: a saved pre-activation is not read again before the backward pass
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <typename T, int CW>
__device__ __forceinline__ void st_elems_nt(void* p, const float (&v)[CW]) {
  if constexpr (sizeof(T) == 4) {
    u32x4_t u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_nontemporal_store(u, reinterpret_cast<u32x4_t*>(p));
  } else if constexpr (CW == 8) {
    u32x4_t u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    __builtin_nontemporal_store(u, reinterpret_cast<u32x4_t*>(p));
  } else {
    u32x2_t u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    __builtin_nontemporal_store(u, reinterpret_cast<u32x2_t*>(p));
  }
}

// SIMPLE = 1: the epilogue has no element-wise side input and no column sums (bias / activation / saved Z / row factors /
// dropout only); SIMPLE = 2: one element-type side input (residual, or the saved pre-activation of a fused act-grad),
// still no C0 rows and no column sums - known at compile time, so the side-input registers, their zero fills and the feature selects vanish
// (PMC: 113 VALU instructions per 8-element item in the general instantiation of a bias+Swish+Z epilogue).
// PHR / NTHR: rows staged per phase and threads of the workgroup (defaults: the 256-thread tiled kernels of gemm.hip).
template <typename T, int OSZ, int TILE_N, int TILE_M, bool EVEC, int SIMPLE = 0, int PHR = 0, int NTHR = 256>
__device__ __forceinline__ void epilogue_phase(const GemmParams& p, const char* smem, const float* side, int ph, int nbase,
                                               int m0, int bz, int split, int t) {
  constexpr int WN = PHR ? PHR : (TILE_M > 128 ? 32 : TILE_N / 2);   // rows staged per phase (phase_rows() of the kernel)
  constexpr int STG_LD = TILE_M * 4 + 16;
  constexpr int CW = 16 / OSZ;                          // output columns per item (16 bytes)
  constexpr int CPR = TILE_M / CW;                      // items per row
  constexpr int RSTEP = NTHR / CPR;                     // rows covered per pass of the workgroup's threads
  constexpr int NIT = WN / RSTEP;                       // items per thread
  constexpr int SW = CW * (int)sizeof(T) / 4;           // 32-bit words of one side-input item (type T)
  constexpr int SWR = OSZ == 4 ? CW : SW;               // ... or of a float32 residual item (SMX_IO_RES_F32: fp32 output only)
  typedef typename std::conditional<OSZ == 4, float, T>::type OutT;
  const smx_epilogue& e = p.e;
  const uint32_t dthresh = p.dthresh;
  const float dscale = p.dscale;
  const uint64_t dseed = dthresh ? epoch_seed(p.e.drop_seed, p.epoch) : 0;
  const int c = (t % CPR) * CW, m = m0 + c, r0 = t / CPR;
  if (m >= p.M) return;
  const float* mkrow = side + TILE_M + ph * WN;
  char* Cb = reinterpret_cast<char*>(p.C) + ((long)bz * p.sC + (long)split * p.sSplit) * OSZ;
  const bool ag = SIMPLE != 1 && (e.flags & SMX_EPI_ACT_GRAD) != 0;     // z is an input: multiply by act'(z)
  const bool c0post = !SIMPLE && (e.flags & SMX_EPI_C0_POST) != 0;
  const bool has_c0 = !SIMPLE && e.c0_mode != SMX_C0_NONE;
  const bool has_mk = e.row_mask != nullptr || e.alpha != 1.f;
  T* Zb = (e.z && !ag) ? reinterpret_cast<T*>(e.z) + (long)bz * p.sC : nullptr;
  // the one side input of element type: the residual, or (ACT_GRAD) the saved pre-activation
  const T* Sb = SIMPLE == 1 ? nullptr
                       : (ag ? reinterpret_cast<const T*>(e.z) : (e.res ? reinterpret_cast<const T*>(e.res) + (long)bz * p.sC : nullptr));
  const long lds_ = ag ? e.ldz : e.ldr;
  // fp32 residual stream: `res` holds float32 (the new stream tensor C is float32 too, OSZ == 4: same 4-column items)
  const bool rf32 = OSZ == 4 && SIMPLE != 1 && !ag && (e.io_flags & SMX_IO_RES_F32) != 0 && e.res != nullptr;
  const float* Sbf = rf32 ? reinterpret_cast<const float*>(e.res) + (long)bz * p.sC : nullptr;

  if constexpr (EVEC) {
    // ---- the element-type side input (residual / saved pre-activation) of ALL the phase's items is requested and
    // waited for before the phase's first store; the (rarer, fp32, twice as wide) C0 rows go in batches of NB items.
    // vmcnt retires in order and counts stores: a load issued after stores can only be consumed once those stores
    // have drained, so every such point is a full write-latency bubble - none of them sits between two stores of a
    // kernel without side inputs, one per phase with a residual, one per batch with C0. ----
    constexpr int NB = NIT < 2 ? NIT : 2;
    uint32_t sw[NIT][SWR];
    if (rf32) {
      if constexpr (OSZ == 4) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int n = min(nbase + r0 + k * RSTEP, p.N - 1);
          ld_words<SWR>(Sbf + (long)n * lds_ + m, sw[k]);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k)
#pragma unroll
          for (int q = 0; q < SWR; ++q) settle(sw[k][q]);
      }
    } else if (Sb) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int n = min(nbase + r0 + k * RSTEP, p.N - 1);
        uint32_t w_[SW];
        ld_words<SW>(Sb + (long)n * lds_ + m, w_);
#pragma unroll
        for (int q = 0; q < SW; ++q) sw[k][q] = w_[q];
#pragma unroll
        for (int q = SW; q < SWR; ++q) sw[k][q] = 0u;
      }
      // settle INSIDE the branch that loads: afterwards no register is a pending load in the compiler's scoreboard on
      // any path, so it cannot place a (conservative) vmcnt wait between the stores below
#pragma unroll
      for (int k = 0; k < NIT; ++k)
#pragma unroll
        for (int q = 0; q < SW; ++q) settle(sw[k][q]);
    } else {
#pragma unroll
      for (int k = 0; k < NIT; ++k)
#pragma unroll
        for (int q = 0; q < SWR; ++q) sw[k][q] = 0u;
    }
#pragma unroll
    for (int kb = 0; kb < NIT; kb += NB) {
    float cv[NB][CW];
    if (has_c0) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int n = min(nbase + r0 + (kb + k) * RSTEP, p.N - 1);
        const float* c0p = e.c0 + c0_row(e, n) * e.ldc0 + m;
#pragma unroll
        for (int q4 = 0; q4 < CW / 4; ++q4) {
          const float4 c4 = *reinterpret_cast<const float4*>(c0p + 4 * q4);
          cv[k][4 * q4] = c4.x; cv[k][4 * q4 + 1] = c4.y; cv[k][4 * q4 + 2] = c4.z; cv[k][4 * q4 + 3] = c4.w;
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int q = 0; q < CW; ++q) settle(cv[k][q]);
    } else {
#pragma unroll
      for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int q = 0; q < CW; ++q) cv[k][q] = 0.f;
    }
    // ---- math + stores: no global load inside, the stores stream out back to back ----
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
      const int k = kb + kk;
      const int r = r0 + k * RSTEP, n = nbase + r;
      if (n >= p.N) continue;
      float v[CW];
#pragma unroll
      for (int q4 = 0; q4 < CW / 4; ++q4) {
        const float4 a4 = *reinterpret_cast<const float4*>(smem + r * STG_LD + (c + 4 * q4) * 4);
        v[4 * q4] = a4.x; v[4 * q4 + 1] = a4.y; v[4 * q4 + 2] = a4.z; v[4 * q4 + 3] = a4.w;
      }
      if (e.bias) {
#pragma unroll
        for (int q4 = 0; q4 < CW / 4; ++q4) {
          const float4 b4 = *reinterpret_cast<const float4*>(side + c + 4 * q4);
          v[4 * q4] += b4.x; v[4 * q4 + 1] += b4.y; v[4 * q4 + 2] += b4.z; v[4 * q4 + 3] += b4.w;
        }
      }
      if (has_c0 && !c0post) {
#pragma unroll
        for (int q = 0; q < CW; ++q) v[q] += cv[kk][q];
      }
      if (ag) {
        float zf[CW];
        { uint32_t w_[SW];
#pragma unroll
          for (int q = 0; q < SW; ++q) w_[q] = sw[k][q];
          unpack_words<T, CW>(w_, zf); }
        switch (e.act) {
          case SMX_ACT_GELU: act_grad_mul_n<SMX_ACT_GELU, CW>(v, zf); break;
          case SMX_ACT_SWISH: act_grad_mul_n<SMX_ACT_SWISH, CW>(v, zf); break;
          case SMX_ACT_LEAKY_RELU: act_grad_mul_n<SMX_ACT_LEAKY_RELU, CW>(v, zf); break;
          case SMX_ACT_RELU: act_grad_mul_n<SMX_ACT_RELU, CW>(v, zf); break;
          default: break;
        }
      } else {
        if (Zb) st_elems_nt<T, CW>(Zb + (long)n * e.ldz + m, v);   // (the saved pre-activation is not read again before the backward pass)
        switch (e.act) {
          case SMX_ACT_GELU: act_fwd_n<SMX_ACT_GELU, CW>(v); break;
          case SMX_ACT_SWISH: act_fwd_n<SMX_ACT_SWISH, CW>(v); break;
          case SMX_ACT_LEAKY_RELU: act_fwd_n<SMX_ACT_LEAKY_RELU, CW>(v); break;
          case SMX_ACT_RELU: act_fwd_n<SMX_ACT_RELU, CW>(v); break;
          default: break;
        }
      }
      if (dthresh && m < p.drop_cols) {                  // fused inverted dropout, mask = f(seed, n * drop_cols + m)
        dropout_apply<CW>(v, dseed, (uint64_t)n * p.drop_cols + m, dthresh, dscale);   // (m and drop_cols are multiples of CW here)
      }
      if (has_mk) {
        const float mk = mkrow[r];
#pragma unroll
        for (int q = 0; q < CW; ++q) v[q] *= mk;
      }
      if (rf32) {
        if constexpr (OSZ == 4) {
#pragma unroll
          for (int q = 0; q < CW; ++q) v[q] += __uint_as_float(sw[k][q]);
        }
      } else if (Sb && !ag) {
        float rf[CW];
        { uint32_t w_[SW];
#pragma unroll
          for (int q = 0; q < SW; ++q) w_[q] = sw[k][q];
          unpack_words<T, CW>(w_, rf); }
#pragma unroll
        for (int q = 0; q < CW; ++q) v[q] += rf[q];
      }
      if (has_c0 && c0post) {
#pragma unroll
        for (int q = 0; q < CW; ++q) v[q] += cv[kk][q];
      }
      if ((!SIMPLE && e.colsum) || (e.flags & SMX_EPI_LN_FWD)) {   // final values back into the item's own staged slot
#pragma unroll
        for (int q4 = 0; q4 < CW / 4; ++q4)
          *reinterpret_cast<float4*>(const_cast<char*>(smem) + r * STG_LD + (c + 4 * q4) * 4) =
              make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
      }
      if (p.nt & 2) st_elems_nt<OutT, CW>(Cb + ((long)n * p.ldc + m) * OSZ, v); else st_elems<OutT, CW>(Cb + ((long)n * p.ldc + m) * OSZ, v);
    }
    }
  } else {
    // ragged / unaligned shapes: one element at a time (rolled loops, run-time activation)
    const int nv = min(CW, p.M - m);
    const float* sf = reinterpret_cast<const float*>(smem);
#pragma unroll 1
    for (int k = 0; k < NIT; ++k) {
      const int r = r0 + k * RSTEP, n = nbase + r;
      if (n >= p.N) continue;
      const float mkv = mkrow[r];
      const float* c0p = has_c0 ? e.c0 + c0_row(e, n) * e.ldc0 + m : nullptr;
#pragma unroll 1
      for (int q = 0; q < nv; ++q) {
        float v = sf[r * (STG_LD / 4) + c + q] + side[c + q];
        if (c0p && !c0post) v += c0p[q];
        if (ag) {
          v *= act_grad(e.act, to_f32(Sb[(long)n * lds_ + m + q]));
        } else {
          if (Zb) Zb[(long)n * e.ldz + m + q] = from_f32<T>(v);
          v = act_fwd(e.act, v);
        }
        if (dthresh && m + q < p.drop_cols) v = dropout_keep(dseed, (uint64_t)n * p.drop_cols + m + q, dthresh) ? v * dscale : 0.f;
        v *= mkv;
        if (rf32) v += Sbf[(long)n * lds_ + m + q];
        else if (Sb && !ag) v += to_f32(Sb[(long)n * lds_ + m + q]);
        if (c0p && c0post) v += c0p[q];
        if (e.colsum) const_cast<float*>(sf)[r * (STG_LD / 4) + c + q] = v;
        if constexpr (OSZ == 4) reinterpret_cast<float*>(Cb)[(long)n * p.ldc + m + q] = v;
        else reinterpret_cast<uint16_t*>(Cb)[(long)n * p.ldc + m + q] = (uint16_t)f32_to_bf16_bits(v);
      }
    }
  }
}

// column sums of a reduce-strided (KS) operand stage: a thread's vectors all cover the SAME columns (256 threads are
// a multiple of the chunks per k row), so it keeps one partial sum per column it owns.  Used by the wgrad GEMM: the
// column sums of dZ are the bias gradient, a by-product of tiles the kernel stages anyway.
template <typename T, int ROWS>
__device__ __forceinline__ void stage_colsum(const uint4 (&reg)[ROWS / 32], float (&cs)[16 / sizeof(T)]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const uint32_t w[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { cs[2 * q] += bf16_bits_to_f32(w[q] & 0xffffu); cs[2 * q + 1] += bf16_bits_to_f32(w[q] >> 16); }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) cs[q] += __uint_as_float(w[q]);
    }
  }
}


// ---- LayerNorm fused into the epilogue of a row-complete tile (TILE_M = W = the LayerNorm width, 256 or 512; 256 threads, 32
// staged rows per phase): thread t owns the 8 columns c = (t % LPR) * 8 of rows r0 + RSTEP k (LPR = W / 8 lanes per row, r0 =
// t / LPR, RSTEP = 256 / LPR), so the 32 lanes of a half wave (W = 256) or the 64 lanes of a wave (W = 512) hold one row and
// a row reduction stays inside the half wave / the wave. ----
// (the sums run on the DPP path, smx_common.h: no LDS round trips)
template <int W>
__device__ __forceinline__ float ln_row_sum(float v) {
  static_assert(W == 256 || W == 512, "LayerNorm-fused tiles: 256 or 512 columns");
  if constexpr (W == 512) return wave_sum_dpp(v);
  else return half_wave_sum_dpp(v);
}

// SMX_EPI_LN_BWD: staged rows = g (gradient of the LayerNorm output).  dX = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
// + res, optional second output alpha2 * D(dX) * mask2, dgamma / dbeta accumulated per thread (folded per tile by the caller).
// (two items at a time: the side inputs of all four would not fit next to the 128 accumulator registers; EXT = the
// variants with a fused activation of the LayerNorm and / or an activation gradient in the second output - a separate
// instantiation, they cost ~20 registers the plain one does not have)
// XF32: the LayerNorm input ln_x is float32 (SMX_IO_LNX_F32, fp32 residual stream); everything else stays dtype T
template <typename T, bool EXT, bool XF32 = false, int W = 256>
__device__ __forceinline__ void epilogue_phase_lnbwd(const GemmParams& p, const char* smem, const float* lng, int nbase0, int t,
                                                     float (&dgam)[8], float (&dbet)[8]) {
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NIT = 16 / RSTEP;   // (W = 256: two rows in flight per thread, W = 512: four)
  constexpr int STG_LD = W * 4 + 16, SW = 8 * (int)sizeof(T) / 4, SWX = XF32 ? 8 : SW;
  constexpr float INVW = 1.f / W;
  typedef typename std::conditional<XF32, float, T>::type XT;
  const smx_epilogue& e = p.e;
  const int c = (t % LPR) * 8, r0 = t / LPR;
  const XT* X = reinterpret_cast<const XT*>(e.ln_x);
  const T* R = reinterpret_cast<const T*>(e.res);
  float gam[8];
#pragma unroll
  for (int q4 = 0; q4 < 2; ++q4) {
    const float4 g4 = *reinterpret_cast<const float4*>(lng + c + 4 * q4);
    gam[4 * q4] = g4.x; gam[4 * q4 + 1] = g4.y; gam[4 * q4 + 2] = g4.z; gam[4 * q4 + 3] = g4.w;
  }
  const int lact = EXT ? e.lnf_act : SMX_ACT_NONE;       // the LayerNorm was followed by a fused activation: g *= act'(LN(x))
  const T* Z2 = EXT ? reinterpret_cast<const T*>(e.z) : nullptr;   // second output: dX2 = alpha2 * D(dX * act'(z)) * mask2
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
  const int nbase = nbase0 + half * 16;
  const char* smh = smem + half * 16 * STG_LD;
  // every side input of the phase is requested (and waited for) before its first store
  uint32_t xw[NIT][SWX], rw[NIT][SW];
  float2 st[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const long n = min(nbase + r0 + k * RSTEP, p.N - 1);
    if constexpr (XF32) {
      const uint4 a_ = *reinterpret_cast<const uint4*>(X + n * e.ln_ldx + c), b_ = *reinterpret_cast<const uint4*>(X + n * e.ln_ldx + c + 4);
      xw[k][0] = a_.x; xw[k][1] = a_.y; xw[k][2] = a_.z; xw[k][3] = a_.w; xw[k][4] = b_.x; xw[k][5] = b_.y; xw[k][6] = b_.z; xw[k][7] = b_.w;
    } else {
      ld_words<SW>(X + n * e.ln_ldx + c, xw[k]);
    }
    st[k] = *reinterpret_cast<const float2*>(e.ln_stats + 2 * n);
    if (R) ld_words<SW>(R + n * e.ldr + c, rw[k]);
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
#pragma unroll
    for (int q = 0; q < SWX; ++q) settle(xw[k][q]);
    settle(st[k].x); settle(st[k].y);
  }
  if (R) {
#pragma unroll
    for (int k = 0; k < NIT; ++k)
#pragma unroll
      for (int q = 0; q < SW; ++q) settle(rw[k][q]);
  } else {
#pragma unroll
    for (int k = 0; k < NIT; ++k)
#pragma unroll
      for (int q = 0; q < SW; ++q) rw[k][q] = 0u;
  }
  const uint32_t thresh2 = e.ln_dx2 ? (uint32_t)((double)e.ln_drop_p2 * 4294967296.0) : 0u;
  const float scale2 = 1.f / (1.f - e.ln_drop_p2);
  const uint64_t seed2 = thresh2 ? epoch_seed(e.ln_drop_seed2, p.epoch) : 0;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int r = r0 + k * RSTEP, n = nbase + r;
    const bool rok = n < p.N;
    float v[8], xh[8], rf[8];
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      const float4 a4 = *reinterpret_cast<const float4*>(smh + r * STG_LD + (c + 4 * q4) * 4);
      v[4 * q4] = a4.x; v[4 * q4 + 1] = a4.y; v[4 * q4 + 2] = a4.z; v[4 * q4 + 3] = a4.w;
    }
    if constexpr (XF32) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xh[q] = __uint_as_float(xw[k][q]);
    } else {
      unpack_words<T, 8>(xw[k], xh);
    }
    unpack_words<T, 8>(rw[k], rf);
    float s1 = 0.f, s2 = 0.f;
    if (EXT && lact != SMX_ACT_NONE) {                   // (uniform; beta sits behind gamma in LDS)
      float ag[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ag[q] = (xh[q] - st[k].x) * st[k].y * gam[q] + lng[W + c + q];
      switch (lact) {
        case SMX_ACT_GELU: act_grad_mul_n<SMX_ACT_GELU, 8>(v, ag); break;
        case SMX_ACT_SWISH: act_grad_mul_n<SMX_ACT_SWISH, 8>(v, ag); break;
        case SMX_ACT_LEAKY_RELU: act_grad_mul_n<SMX_ACT_LEAKY_RELU, 8>(v, ag); break;
        case SMX_ACT_RELU: act_grad_mul_n<SMX_ACT_RELU, 8>(v, ag); break;
        default: break;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      xh[q] = (xh[q] - st[k].x) * st[k].y;
      const float g = rok ? v[q] : 0.f;
      dgam[q] += g * xh[q];
      dbet[q] += g;
      v[q] = g * gam[q];
      s1 += v[q];
      s2 += v[q] * xh[q];
    }
    s1 = ln_row_sum<W>(s1) * INVW;
    s2 = ln_row_sum<W>(s2) * INVW;
    if (!rok) continue;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = st[k].y * (v[q] - s1 - xh[q] * s2) + rf[q];
    st_elems<T, 8>(reinterpret_cast<T*>(p.C) + (long)n * p.ldc + c, v);
    if (e.ln_dx2) {                                      // (uniform)
      const float mk = (e.ln_mask2 ? (e.ln_mask2[n] ? 1.f : 0.f) : 1.f) * e.ln_alpha2;
      if (EXT && Z2) {                                   // (a load behind the dX store: one drain per item, second output only)
        uint32_t zw[SW];
        float zf[8];
        ld_words<SW>(Z2 + (long)n * e.ldz + c, zw);
        unpack_words<T, 8>(zw, zf);
        switch (e.act) {
          case SMX_ACT_GELU: act_grad_mul_n<SMX_ACT_GELU, 8>(v, zf); break;
          case SMX_ACT_SWISH: act_grad_mul_n<SMX_ACT_SWISH, 8>(v, zf); break;
          case SMX_ACT_LEAKY_RELU: act_grad_mul_n<SMX_ACT_LEAKY_RELU, 8>(v, zf); break;
          case SMX_ACT_RELU: act_grad_mul_n<SMX_ACT_RELU, 8>(v, zf); break;
          default: break;
        }
      }
      if (thresh2) dropout_apply<8>(v, seed2, (uint64_t)n * W + c, thresh2, scale2);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] *= mk;
      st_elems<T, 8>(reinterpret_cast<T*>(e.ln_dx2) + (long)n * e.ln_lddx2 + c, v);
    }
  }
  }
}

// ---- the same, split into REQUEST and CONSUME: the side inputs of 16 rows (a half phase: rows r0 + RSTEP k, k < NH, of the thread's
// half wave / wave) - LayerNorm input, statistics, residual gradient, the second output's mask byte and saved pre-activation - go
// into ONE register set; the first half of a phase is requested BEFORE the accumulator dump of that phase, so its round trip runs
// under the dump and its barriers, and the mask / pre-activation of the second output are no longer loaded behind the dX store.
template <typename T, bool EXT, bool XF32, int W>
struct LnBwdIn {
  static constexpr int NH = 16 / (256 / (W / 8));         // rows of a half phase per thread (W = 256: 2, W = 512: 4)
  uint32_t xw[NH][XF32 ? 8 : 4], rw[NH][4], zw[EXT ? NH : 1][4], mk[NH];
  float2 st[NH];
};
template <typename T, bool EXT, bool XF32, int W>
__device__ __forceinline__ void ln_bwd_request(const GemmParams& p, int nbase, int t, LnBwdIn<T, EXT, XF32, W>& in) {
  typedef typename std::conditional<XF32, float, T>::type XT;
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NH = 16 / RSTEP;
  const smx_epilogue& e = p.e;
  const int c = (t % LPR) * 8, r0 = t / LPR;
  const XT* X = reinterpret_cast<const XT*>(e.ln_x);
  const T* R = reinterpret_cast<const T*>(e.res);
  const T* Z2 = EXT ? reinterpret_cast<const T*>(e.z) : nullptr;
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const long n = min(nbase + r0 + RSTEP * k, p.N - 1);
    if constexpr (XF32) {
      const uint4 a_ = *reinterpret_cast<const uint4*>(X + n * e.ln_ldx + c), b_ = *reinterpret_cast<const uint4*>(X + n * e.ln_ldx + c + 4);
      in.xw[k][0] = a_.x; in.xw[k][1] = a_.y; in.xw[k][2] = a_.z; in.xw[k][3] = a_.w;
      in.xw[k][4] = b_.x; in.xw[k][5] = b_.y; in.xw[k][6] = b_.z; in.xw[k][7] = b_.w;
    } else {
      ld_words<4>(X + n * e.ln_ldx + c, in.xw[k]);
    }
    in.st[k] = *reinterpret_cast<const float2*>(e.ln_stats + 2 * n);
    if (R) ld_words<4>(R + n * e.ldr + c, in.rw[k]);
    if (e.ln_dx2 && e.ln_mask2) in.mk[k] = e.ln_mask2[n];
    if constexpr (EXT) {
      if (Z2 && e.ln_dx2) ld_words<4>(Z2 + n * e.ldz + c, in.zw[k]);
    }
  }
}
template <typename T, bool EXT, bool XF32, int W>
__device__ __forceinline__ void ln_bwd_half(const GemmParams& p, const char* smh, const float* lng, int nbase, int t,
                                            LnBwdIn<T, EXT, XF32, W>& in, float (&dgam)[8], float (&dbet)[8]) {
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NH = 16 / RSTEP, STG_LD = W * 4 + 16, SWX = XF32 ? 8 : 4;
  constexpr float INVW = 1.f / W;
  const smx_epilogue& e = p.e;
  const int c = (t % LPR) * 8, r0 = t / LPR;
  const bool hasR = e.res != nullptr, hasZ2 = EXT && e.z != nullptr && e.ln_dx2 != nullptr, hasM2 = e.ln_dx2 != nullptr && e.ln_mask2 != nullptr;
  float gam[8];
#pragma unroll
  for (int q4 = 0; q4 < 2; ++q4) {
    const float4 g4 = *reinterpret_cast<const float4*>(lng + c + 4 * q4);
    gam[4 * q4] = g4.x; gam[4 * q4 + 1] = g4.y; gam[4 * q4 + 2] = g4.z; gam[4 * q4 + 3] = g4.w;
  }
  const int lact = EXT ? e.lnf_act : SMX_ACT_NONE;       // the LayerNorm was followed by a fused activation: g *= act'(LN(x))
  // every (uniform) branch that loaded settles its own registers: no pending load on any path afterwards
#pragma unroll
  for (int k = 0; k < NH; ++k) {
#pragma unroll
    for (int q = 0; q < SWX; ++q) settle(in.xw[k][q]);
    settle(in.st[k].x); settle(in.st[k].y);
  }
  if (hasR) {
#pragma unroll
    for (int k = 0; k < NH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) settle(in.rw[k][q]);
  } else {
#pragma unroll
    for (int k = 0; k < NH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) in.rw[k][q] = 0u;
  }
  if (hasM2) {
#pragma unroll
    for (int k = 0; k < NH; ++k) settle(in.mk[k]);
  } else {
#pragma unroll
    for (int k = 0; k < NH; ++k) in.mk[k] = 1u;
  }
  if constexpr (EXT) {
    if (hasZ2) {
#pragma unroll
      for (int k = 0; k < NH; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) settle(in.zw[k][q]);
    }
  }
  const uint32_t thresh2 = e.ln_dx2 ? (uint32_t)((double)e.ln_drop_p2 * 4294967296.0) : 0u;
  const float scale2 = 1.f / (1.f - e.ln_drop_p2);
  const uint64_t seed2 = thresh2 ? epoch_seed(e.ln_drop_seed2, p.epoch) : 0;
  // two rows at a time (four independent reduction chains): with four the 2 x 60 request registers + 64 of v / xhat spill
#pragma unroll
  for (int kb = 0; kb < NH; kb += 2) {
  float v[2][8], xh[2][8], s1[2], s2[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = kb + kk;
    const int r = r0 + RSTEP * k;
    const bool rok = nbase + r < p.N;
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      const float4 a4 = *reinterpret_cast<const float4*>(smh + r * STG_LD + (c + 4 * q4) * 4);
      v[kk][4 * q4] = a4.x; v[kk][4 * q4 + 1] = a4.y; v[kk][4 * q4 + 2] = a4.z; v[kk][4 * q4 + 3] = a4.w;
    }
    if constexpr (XF32) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xh[kk][q] = __uint_as_float(in.xw[k][q]);
    } else {
      unpack_words<T, 8>(in.xw[k], xh[kk]);
    }
    if (EXT && lact != SMX_ACT_NONE) {                   // (uniform; beta sits behind gamma in LDS)
      float ag[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ag[q] = (xh[kk][q] - in.st[k].x) * in.st[k].y * gam[q] + lng[W + c + q];
      switch (lact) {
        case SMX_ACT_GELU: act_grad_mul_n<SMX_ACT_GELU, 8>(v[kk], ag); break;
        case SMX_ACT_SWISH: act_grad_mul_n<SMX_ACT_SWISH, 8>(v[kk], ag); break;
        case SMX_ACT_LEAKY_RELU: act_grad_mul_n<SMX_ACT_LEAKY_RELU, 8>(v[kk], ag); break;
        case SMX_ACT_RELU: act_grad_mul_n<SMX_ACT_RELU, 8>(v[kk], ag); break;
        default: break;
      }
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      xh[kk][q] = (xh[kk][q] - in.st[k].x) * in.st[k].y;
      const float g = rok ? v[kk][q] : 0.f;
      dgam[q] += g * xh[kk][q];
      dbet[q] += g;
      v[kk][q] = g * gam[q];
      a1 += v[kk][q];
      a2 += v[kk][q] * xh[kk][q];
    }
    s1[kk] = a1; s2[kk] = a2;
  }
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) { s1[kk] = ln_row_sum<W>(s1[kk]) * INVW; s2[kk] = ln_row_sum<W>(s2[kk]) * INVW; }   // four independent chains
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = kb + kk;
    const long n = nbase + r0 + RSTEP * k;
    if (n >= p.N) continue;
    float rf[8];
    unpack_words<T, 8>(in.rw[k], rf);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[kk][q] = in.st[k].y * (v[kk][q] - s1[kk] - xh[kk][q] * s2[kk]) + rf[q];
    st_elems<T, 8>(reinterpret_cast<T*>(p.C) + n * p.ldc + c, v[kk]);
    if (e.ln_dx2) {                                      // (uniform)
      const float mk = (in.mk[k] ? 1.f : 0.f) * e.ln_alpha2;
      if constexpr (EXT) {
        if (hasZ2) {
          float zf[8];
          unpack_words<T, 8>(in.zw[k], zf);
          switch (e.act) {
            case SMX_ACT_GELU: act_grad_mul_n<SMX_ACT_GELU, 8>(v[kk], zf); break;
            case SMX_ACT_SWISH: act_grad_mul_n<SMX_ACT_SWISH, 8>(v[kk], zf); break;
            case SMX_ACT_LEAKY_RELU: act_grad_mul_n<SMX_ACT_LEAKY_RELU, 8>(v[kk], zf); break;
            case SMX_ACT_RELU: act_grad_mul_n<SMX_ACT_RELU, 8>(v[kk], zf); break;
            default: break;
          }
        }
      }
      if (thresh2) dropout_apply<8>(v[kk], seed2, (uint64_t)n * W + c, thresh2, scale2);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[kk][q] *= mk;
      st_elems<T, 8>(reinterpret_cast<T*>(e.ln_dx2) + n * e.ln_lddx2 + c, v[kk]);
    }
  }
  }
}

// SMX_EPI_LN_FWD: the ordinary epilogue_phase has written the finished output values back to their staged slots (each
// thread re-reads its own items: no barrier); lnf_y = act(LN(row) * gamma + beta), lnf_stats = (mean, rstd).
template <typename T, int W = 256>
__device__ __forceinline__ void epilogue_phase_lnfwd(const GemmParams& p, const char* smem, const float* lng, int nbase, int t) {
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NIT = 32 / RSTEP, STG_LD = W * 4 + 16;
  constexpr float INVW = 1.f / W;
  const smx_epilogue& e = p.e;
  const int c = (t % LPR) * 8, r0 = t / LPR;
  float gam[8], bet[8];
#pragma unroll
  for (int q4 = 0; q4 < 2; ++q4) {
    const float4 g4 = *reinterpret_cast<const float4*>(lng + c + 4 * q4), b4 = *reinterpret_cast<const float4*>(lng + W + c + 4 * q4);
    gam[4 * q4] = g4.x; gam[4 * q4 + 1] = g4.y; gam[4 * q4 + 2] = g4.z; gam[4 * q4 + 3] = g4.w;
    bet[4 * q4] = b4.x; bet[4 * q4 + 1] = b4.y; bet[4 * q4 + 2] = b4.z; bet[4 * q4 + 3] = b4.w;
  }
  // BR rows at a time: their reductions are independent dependency chains (one workgroup per CU on the 512-wide tile: nobody else
  // hides them); the 256-wide kernels, two per CU at the 256-register budget, keep one row live
  constexpr int BR = W == 512 ? 4 : 1;
#pragma unroll
  for (int kb = 0; kb < NIT; kb += BR) {
    float v[BR][8], mean[BR], rstd[BR];
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      const int r = r0 + (kb + k) * RSTEP;
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        const float4 a4 = *reinterpret_cast<const float4*>(smem + r * STG_LD + (c + 4 * q4) * 4);
        v[k][4 * q4] = a4.x; v[k][4 * q4 + 1] = a4.y; v[k][4 * q4 + 2] = a4.z; v[k][4 * q4 + 3] = a4.w;
      }
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[k][q];
      mean[k] = ln_row_sum<W>(s) * INVW;
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[k][q] -= mean[k]; qq += v[k][q] * v[k][q]; }
      rstd[k] = rsqrtf(ln_row_sum<W>(qq) * INVW + e.lnf_eps);
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      const int n = nbase + r0 + (kb + k) * RSTEP;
      if (n >= p.N) continue;
      float y[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = v[k][q] * rstd[k] * gam[q] + bet[q];
      switch (e.lnf_act) {
        case SMX_ACT_GELU: act_fwd_n<SMX_ACT_GELU, 8>(y); break;
        case SMX_ACT_SWISH: act_fwd_n<SMX_ACT_SWISH, 8>(y); break;
        case SMX_ACT_LEAKY_RELU: act_fwd_n<SMX_ACT_LEAKY_RELU, 8>(y); break;
        case SMX_ACT_RELU: act_fwd_n<SMX_ACT_RELU, 8>(y); break;
        default: break;
      }
      if (e.io_flags & SMX_IO_LNFY_F32) {                  // (uniform) the LayerNorm output IS the fp32 residual stream (norm2)
        float* yp = reinterpret_cast<float*>(e.lnf_y) + (long)n * e.lnf_ldy + c;
        *reinterpret_cast<float4*>(yp) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(yp + 4) = make_float4(y[4], y[5], y[6], y[7]);
      } else {
        st_elems<T, 8>(reinterpret_cast<T*>(e.lnf_y) + (long)n * e.lnf_ldy + c, y);
      }
      if (e.lnf_stats && (t % LPR) == 0) *reinterpret_cast<float2*>(e.lnf_stats + 2 * (long)n) = make_float2(mean[k], rstd[k]);
    }
  }
}

// ---- LayerNorm forward on the float32 residual stream (SMX_IO_RES_F32, float32 C) in ONE pass per phase (row-complete tiles of
// W = 256 or 512 columns) -------------------------------------------------------------------------------------------------------
// The generic pair above (epilogue_phase with 4-column float32 items, write-back to the staged slots, barrier, epilogue_phase_lnfwd
// with 8-column items) costs a second LDS round trip and a barrier per phase.  Here thread t owns the 8 columns c = (t % LPR) * 8 of
// rows r0 + RSTEP k (LPR = W / 8 lanes per row: a half wave or a wave = a row) for the bias / activation / dropout / mask / residual
// part AND the LayerNorm: the finished values never leave the registers, the row statistics are two DPP sums per row.  The residual
// rows of the phase are requested BEFORE the accumulator dump (ln1p_request_res), so their round trip runs under the dump and its
// two barriers.  Eligibility (checked by the kernel, else the generic pair): SIMPLE == 2 epilogue (bias / activation / saved Z /
// dropout / row factors / one residual), float32 residual and output.
template <int W>
__device__ __forceinline__ void ln1p_request_res(const GemmParams& p, int nbase, int t, uint32_t (&rw)[W / 64][8]) {
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NIT = 32 / RSTEP;
  const float* R = reinterpret_cast<const float*>(p.e.res);
  const int c = (t % LPR) * 8, r0 = t / LPR;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const long n = min(nbase + r0 + RSTEP * k, p.N - 1);
    const uint4 a_ = *reinterpret_cast<const uint4*>(R + n * p.e.ldr + c), b_ = *reinterpret_cast<const uint4*>(R + n * p.e.ldr + c + 4);
    rw[k][0] = a_.x; rw[k][1] = a_.y; rw[k][2] = a_.z; rw[k][3] = a_.w; rw[k][4] = b_.x; rw[k][5] = b_.y; rw[k][6] = b_.z; rw[k][7] = b_.w;
  }
}
// lnst: [2][128][2] floats of LDS - the tile's (mean, rstd) pairs (of the LayerNorm and of the optional second one), written out as
// contiguous 1 KB blocks by ln1p_store_stats after the last phase
template <typename T, int W>
__device__ __forceinline__ void epilogue_phase_ln1p(const GemmParams& p, const char* smem, const float* side, const float* lng, float* lnst,
                                                    const float* lng2, int ph, int nbase, int t, uint32_t (&rw)[W / 64][8]) {
  constexpr int LPR = W / 8, RSTEP = 256 / LPR, NIT = 32 / RSTEP, STG_LD = W * 4 + 16;
  constexpr int BR = W == 512 ? 4 : 2;                   // rows in flight (the 256-wide kernels run two per CU at the 256-register budget)
  constexpr float INVW = 1.f / W;
  const smx_epilogue& e = p.e;
  const int c = (t % LPR) * 8, r0 = t / LPR;
  const uint32_t dthresh = p.dthresh;
  const float dscale = p.dscale;
  const uint64_t dseed = dthresh ? epoch_seed(e.drop_seed, p.epoch) : 0;
  const bool has_mk = e.row_mask != nullptr || e.alpha != 1.f;
  const bool do_drop = dthresh && c < p.drop_cols;         // (drop_cols is a multiple of 8: an item is wholly inside or outside)
  const float* mkrow = side + W + ph * 32;
  const bool yf32 = (e.io_flags & SMX_IO_LNFY_F32) != 0;  // the LayerNorm output IS the fp32 residual stream (norm2)
  // row pointers of this thread's first row; every further row is RSTEP rows on (one 64-bit add each, nothing recomputed per row)
  const long row0 = nbase + r0;
  float* cp = reinterpret_cast<float*>(p.C) + row0 * p.ldc + c;
  T* zp = e.z ? reinterpret_cast<T*>(e.z) + row0 * e.ldz + c : nullptr;
  char* yp = reinterpret_cast<char*>(e.lnf_y) + (row0 * e.lnf_ldy + c) * (yf32 ? 4 : (long)sizeof(T));
  const long cstep = RSTEP * p.ldc, zstep = RSTEP * e.ldz, ystep = RSTEP * e.lnf_ldy * (yf32 ? 4 : (long)sizeof(T));
  float gam[8], bet[8], bia[8];
#pragma unroll
  for (int q4 = 0; q4 < 2; ++q4) {
    const float4 g4 = *reinterpret_cast<const float4*>(lng + c + 4 * q4), b4 = *reinterpret_cast<const float4*>(lng + W + c + 4 * q4);
    const float4 s4 = *reinterpret_cast<const float4*>(side + c + 4 * q4);
    gam[4 * q4] = g4.x; gam[4 * q4 + 1] = g4.y; gam[4 * q4 + 2] = g4.z; gam[4 * q4 + 3] = g4.w;
    bet[4 * q4] = b4.x; bet[4 * q4 + 1] = b4.y; bet[4 * q4 + 2] = b4.z; bet[4 * q4 + 3] = b4.w;
    bia[4 * q4] = s4.x; bia[4 * q4 + 1] = s4.y; bia[4 * q4 + 2] = s4.z; bia[4 * q4 + 3] = s4.w;
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) settle(rw[k][q]);         // the phase's residual rows have landed: no load below, the stores stream
#pragma unroll
  for (int kb = 0; kb < NIT; kb += BR) {
    float v[BR][8], mean[BR], rstd[BR];
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      const int r = r0 + RSTEP * (kb + k);
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        const float4 a4 = *reinterpret_cast<const float4*>(smem + r * STG_LD + (c + 4 * q4) * 4);
        v[k][4 * q4] = a4.x + bia[4 * q4]; v[k][4 * q4 + 1] = a4.y + bia[4 * q4 + 1];      // (bias: zeros when there is none)
        v[k][4 * q4 + 2] = a4.z + bia[4 * q4 + 2]; v[k][4 * q4 + 3] = a4.w + bia[4 * q4 + 3];
      }
    }
    if (zp) {                                              // (uniform) saved pre-activation
#pragma unroll
      for (int k = 0; k < BR; ++k)
        if (nbase + r0 + RSTEP * (kb + k) < p.N) st_elems_nt<T, 8>(zp + (kb + k) * zstep, v[k]);
    }
    switch (e.act) {                                       // ONE uniform switch per batch of rows
      case SMX_ACT_GELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_GELU, 8>(v[k]);
        break;
      case SMX_ACT_SWISH:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_SWISH, 8>(v[k]);
        break;
      case SMX_ACT_LEAKY_RELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_LEAKY_RELU, 8>(v[k]);
        break;
      case SMX_ACT_RELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_RELU, 8>(v[k]);
        break;
      default: break;
    }
    if (do_drop) {
#pragma unroll
      for (int k = 0; k < BR; ++k)
        dropout_apply<8>(v[k], dseed, (uint64_t)(nbase + r0 + RSTEP * (kb + k)) * p.drop_cols + c, dthresh, dscale);
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      const float mk = has_mk ? mkrow[r0 + RSTEP * (kb + k)] : 1.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[k][q] = v[k][q] * mk + __uint_as_float(rw[kb + k][q]);
      if (nbase + r0 + RSTEP * (kb + k) < p.N) {               // C: the new stream tensor, streamed past the caches (the next kernel reads lnf_y)
        float* cq = cp + (kb + k) * cstep;
        st_elems_nt<float, 4>(cq, reinterpret_cast<const float(&)[4]>(v[k][0]));
        st_elems_nt<float, 4>(cq + 4, reinterpret_cast<const float(&)[4]>(v[k][4]));
      }
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      float s_ = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s_ += v[k][q];
      mean[k] = ln_row_sum<W>(s_) * INVW;
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[k][q] -= mean[k]; qq += v[k][q] * v[k][q]; }
      rstd[k] = rsqrtf(ln_row_sum<W>(qq) * INVW + e.lnf_eps);
    }
#pragma unroll
    for (int k = 0; k < BR; ++k)
#pragma unroll
      for (int q = 0; q < 8; ++q) v[k][q] = v[k][q] * rstd[k] * gam[q] + bet[q];
    switch (e.lnf_act) {
      case SMX_ACT_GELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_GELU, 8>(v[k]);
        break;
      case SMX_ACT_SWISH:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_SWISH, 8>(v[k]);
        break;
      case SMX_ACT_LEAKY_RELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_LEAKY_RELU, 8>(v[k]);
        break;
      case SMX_ACT_RELU:
#pragma unroll
        for (int k = 0; k < BR; ++k) act_fwd_n<SMX_ACT_RELU, 8>(v[k]);
        break;
      default: break;
    }
#pragma unroll
    for (int k = 0; k < BR; ++k) {
      const int r = r0 + RSTEP * (kb + k);
      if ((t % LPR) == 0) *reinterpret_cast<float2*>(lnst + 2 * (ph * 32 + r)) = make_float2(mean[k], rstd[k]);
      if (nbase + r >= p.N) continue;
      char* yq = yp + (kb + k) * ystep;
      if (yf32) {                                          // (uniform)
        *reinterpret_cast<float4*>(yq) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
        *reinterpret_cast<float4*>(yq + 16) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
      } else {
        st_elems<T, 8>(yq, v[k]);
      }
    }
    if (e.lnf2_y) {
      // (uniform) the SECOND LayerNorm, of the values just stored (a layer's norm2 -> the next layer's first LayerNorm): they are
      // still in the registers, two more sums per row
      float g2[8], b2[8];
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        const float4 g4 = *reinterpret_cast<const float4*>(lng2 + c + 4 * q4), b4 = *reinterpret_cast<const float4*>(lng2 + W + c + 4 * q4);
        g2[4 * q4] = g4.x; g2[4 * q4 + 1] = g4.y; g2[4 * q4 + 2] = g4.z; g2[4 * q4 + 3] = g4.w;
        b2[4 * q4] = b4.x; b2[4 * q4 + 1] = b4.y; b2[4 * q4 + 2] = b4.z; b2[4 * q4 + 3] = b4.w;
      }
#pragma unroll
      for (int k = 0; k < BR; ++k) {
        float s_ = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s_ += v[k][q];
        mean[k] = ln_row_sum<W>(s_) * INVW;
      }
#pragma unroll
      for (int k = 0; k < BR; ++k) {
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[k][q] -= mean[k]; qq += v[k][q] * v[k][q]; }
        rstd[k] = rsqrtf(ln_row_sum<W>(qq) * INVW + e.lnf2_eps);
      }
#pragma unroll
      for (int k = 0; k < BR; ++k) {
        const int r = r0 + RSTEP * (kb + k);
        if ((t % LPR) == 0) *reinterpret_cast<float2*>(lnst + 256 + 2 * (ph * 32 + r)) = make_float2(mean[k], rstd[k]);
        if (nbase + r >= p.N) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[k][q] = v[k][q] * rstd[k] * g2[q] + b2[q];
        st_elems<T, 8>(reinterpret_cast<T*>(e.lnf2_y) + (long)(nbase + r) * e.lnf2_ldy + c, v[k]);
      }
    }
  }
}
// the tile's statistics: rows [n0, n0 + 128) as one contiguous block (call after a barrier behind the last phase)
__device__ __forceinline__ void ln1p_store_stats(const GemmParams& p, const float* lnst, int n0, int t, int rows = 128) {
  if (p.e.lnf_stats && t < rows && n0 + t < p.N)
    *reinterpret_cast<float2*>(p.e.lnf_stats + 2 * (long)(n0 + t)) = *reinterpret_cast<const float2*>(lnst + 2 * t);
  if (p.e.lnf2_y && p.e.lnf2_stats && t >= 128 && t - 128 < rows && n0 + t - 128 < p.N)
    *reinterpret_cast<float2*>(p.e.lnf2_stats + 2 * (long)(n0 + t - 128)) = *reinterpret_cast<const float2*>(lnst + 2 * t);
}

// LDS-DMA issue of one 1 KB piece (global_load_lds_dwordx4: lane i lands at lds_dst + 16 i; M0 carries the wave-uniform
// LDS base and is compiler-reserved, so it is saved / restored inside the statement).  hipcc does not count this
// load: the kernel waits for it with explicit s_waitcnt vmcnt(N).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// fragment of a reduce-strided stage with 256-byte k rows (128 columns), granule XOR 4 * (k & 3): see gemm_tn_dma_kernel
__device__ __forceinline__ bf16x8 frag_tr_swz(const char* lds, int r, int kk, int hi) {
  typedef short short4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const int lane = (r & 31) | (hi << 5);
  const int li = lane & 15, g1 = (lane >> 4) & 1;
  const int k = kk * 16 + hi * 8 + (li >> 2);           // (k & 3) == li >> 2; row k + 4 has the same swizzle
  const int c = (r - (r & 31)) + g1 * 16 + (li & 3) * 4;
  const char* p0 = lds + k * 256 + ((((c >> 3) ^ ((li >> 2) << 2)) << 4) + (c & 7) * 2);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * 256));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return __builtin_bit_cast(bf16x8, make_uint4(ua.x, ua.y, ub.x, ub.y));
}

// fragment of a reduce-strided stage with RB-byte k rows (RB / 2 columns, no pad), 16-byte granule g of row k stored at granule
// g ^ (4 * (k & 3)) - the image of wgrad_group.hip: the four k rows a 16-lane group of ds_read_b64_tr_b16 touches land on four
// disjoint bank quadruples whatever the row length (RB a multiple of 256 B).  cbase = first column of the 32-column fragment.
template <int RB>
__device__ __forceinline__ bf16x8 frag_tr_swz_rb(const char* lds, int cbase, int lane, int kk) {
  typedef short short4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) short4_t* lds_s4;
  const int li = lane & 15, g1 = (lane >> 4) & 1, hi = lane >> 5;
  const int k = kk * 16 + hi * 8 + (li >> 2);           // (k & 3) == li >> 2; row k + 4 has the same swizzle
  const int c = cbase + g1 * 16 + (li & 3) * 4;
  const char* p0 = lds + k * RB + ((((c >> 3) ^ ((li >> 2) << 2)) << 4) + (c & 7) * 2);
  const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0));
  const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p0 + 4 * RB));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return __builtin_bit_cast(bf16x8, make_uint4(ua.x, ua.y, ub.x, ub.y));
}
}  // namespace smx
