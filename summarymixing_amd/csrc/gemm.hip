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
};

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> { static constexpr int BK = 64; static constexpr int VPT = 8; };
template <> struct ElemTraits<float>  { static constexpr int BK = 32; static constexpr int VPT = 4; };

// ---- guarded 16-byte fetch of VPT consecutive elements -------------------------------------------------
template <typename T, bool VEC>
__device__ __forceinline__ uint4 ld_contig(const T* p, int nvalid) {
  constexpr int VPT = ElemTraits<T>::VPT;
  if (VEC && nvalid >= VPT) return *reinterpret_cast<const uint4*>(p);
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

// ---- stage one operand tile (ROWS x BK) : global -> registers ------------------------------------------
// KC: element (row r, reduce k) at base[r*ld + k];  KS: at base[k*ld + r]
template <typename T, bool KC, int ROWS, bool VEC>
__device__ __forceinline__ void stage_load(uint4 (&reg)[4], const T* base, long ld, int row0, int rows_total,
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
      reg[i] = ld_contig<T, VEC>(base + (long)rg * ld + kg, nv);
    }
  } else if constexpr (sizeof(T) == 2 && !KC) {
    constexpr int RC = ROWS / 8;           // 16-byte row chunks per k row
    constexpr int ITEMS = 16 * RC;         // (k quad, row chunk)
    if (t < ITEMS) {
      int rc = t % RC, kq = t / RC;
      int rg = row0 + rc * 8;
      int nv = rows_total - rg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kg = k0 + kq * 4 + j;
        reg[j] = ld_contig<T, VEC>(base + (long)kg * ld + rg, kg < k1 ? nv : 0);
      }
    }
  } else if constexpr (sizeof(T) == 4 && KC) {
    constexpr int NV = ROWS / 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, k4 = v & 7;
      int rg = row0 + row, kg = k0 + k4 * 4;
      int nv = (rg < rows_total) ? (k1 - kg) : 0;
      reg[i] = ld_contig<T, VEC>(base + (long)rg * ld + kg, nv);
    }
  } else {
    constexpr int NV = ROWS / 32;
    constexpr int R4 = ROWS / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, r4 = v % R4, k = v / R4;
      int rg = row0 + r4 * 4, kg = k0 + k;
      reg[i] = ld_contig<T, VEC>(base + (long)kg * ld + rg, kg < k1 ? rows_total - rg : 0);
    }
  }
  (void)BK; (void)VPT;
}

// ---- registers -> LDS image -----------------------------------------------------------------------------
template <typename T, bool KC, int ROWS>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[4], char* lds, int t) {
  if constexpr (sizeof(T) == 2 && KC) {
    constexpr int NV = ROWS / 32;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = t + 256 * i, row = v >> 3, c = v & 7;
      *reinterpret_cast<uint4*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = reg[i];
    }
  } else if constexpr (sizeof(T) == 2 && !KC) {
    constexpr int RC = ROWS / 8;
    constexpr int ITEMS = 16 * RC;
    if (t < ITEMS) {
      int rc = t % RC, kq = t / RC;
      const uint32_t* g0 = reinterpret_cast<const uint32_t*>(&reg[0]);
      const uint32_t* g1 = reinterpret_cast<const uint32_t*>(&reg[1]);
      const uint32_t* g2 = reinterpret_cast<const uint32_t*>(&reg[2]);
      const uint32_t* g3 = reinterpret_cast<const uint32_t*>(&reg[3]);
      int sw = (rc >> 1) & 3;
      char* basep = lds + kq * (ROWS * 8) + (rc >> 2) * 256;
#pragma unroll
      for (int gr = 0; gr < 4; ++gr) {  // granule = rows (2gr, 2gr+1) of this 8-row chunk, 4 k each
        uint32_t w0 = g0[gr], w1 = g1[gr], w2 = g2[gr], w3 = g3[gr];
        uint4 o;
        o.x = (w0 & 0xffffu) | (w1 << 16);          // even row: k0,k1
        o.y = (w2 & 0xffffu) | (w3 << 16);          //           k2,k3
        o.z = (w0 >> 16) | (w1 & 0xffff0000u);      // odd row:  k0,k1
        o.w = (w2 >> 16) | (w3 & 0xffff0000u);      //           k2,k3
        int pg = ((rc & 3) << 2) | (gr ^ sw);
        *reinterpret_cast<uint4*>(basep + (pg << 4)) = o;
      }
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

template <typename T, int ROWS>
constexpr int lds_bytes() {
  return sizeof(T) == 2 ? ROWS * 128 : 32 * (ROWS + 4) * 4;
}

// ---- fragment reads ---------------------------------------------------------------------------------------
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 frag_bf16(const char* lds, int r, int kk, int hi) {
  if constexpr (KC) {
    int c = kk * 2 + hi;
    uint4 v = *reinterpret_cast<const uint4*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    int kq = kk * 4 + hi * 2;
    const char* p = lds + kq * (ROWS * 8) + (r >> 5) * 256 + ((((r & 31) >> 1) ^ ((r >> 4) & 3)) << 4) + (r & 1) * 8;
    uint2 lo = *reinterpret_cast<const uint2*>(p);
    uint2 hi2 = *reinterpret_cast<const uint2*>(p + ROWS * 8);
    uint4 v = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
    return __builtin_bit_cast(bf16x8, v);
  }
}

__device__ __forceinline__ long c0_row(const smx_epilogue& e, int n) {
  if (e.c0_mode == SMX_C0_GROUP) return n / e.c0_div;
  if (e.c0_mode == SMX_C0_MOD) return n % e.c0_div;
  return n;
}

// ---- the kernel ---------------------------------------------------------------------------------------------
template <typename T, bool A_KC, bool B_KC, int TILE_N, int TILE_M, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BK = ElemTraits<T>::BK;
  constexpr int WN = TILE_N / 2, WM = TILE_M / 2;
  constexpr int FN = WN / 32, FM = WM / 32;
  constexpr int A_BYTES = lds_bytes<T, TILE_N>();
  constexpr int B_BYTES = lds_bytes<T, TILE_M>();
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + B_BYTES];
  char* As = smem;
  char* Bs = smem + A_BYTES;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wm = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware tile mapping: consecutive remapped ids walk the M tiles of one row panel
  int ntiles = p.tiles_n * p.tiles_m;
  int bid = blockIdx.x;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid / p.tiles_m, tile_m = bid % p.tiles_m;
  const int n0 = tile_n * TILE_N, m0 = tile_m * TILE_M;
  const int bz = blockIdx.y / p.splits, split = blockIdx.y % p.splits;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const T* A = reinterpret_cast<const T*>(p.A) + (long)bz * p.sA;
  const T* B = reinterpret_cast<const T*>(p.B) + (long)bz * p.sB;

  f32x16 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint4 ra[4], rb[4];
  if (kbeg < kend) {
    stage_load<T, A_KC, TILE_N, VEC>(ra, A, p.lda, n0, p.N, kbeg, kend, t);
    stage_load<T, B_KC, TILE_M, VEC>(rb, B, p.ldb, m0, p.M, kbeg, kend, t);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    stage_store<T, A_KC, TILE_N>(ra, As, t);
    stage_store<T, B_KC, TILE_M>(rb, Bs, t);
    __syncthreads();
    if (k0 + BK < kend) {
      stage_load<T, A_KC, TILE_N, VEC>(ra, A, p.lda, n0, p.N, k0 + BK, kend, t);
      stage_load<T, B_KC, TILE_M, VEC>(rb, B, p.ldb, m0, p.M, k0 + BK, kend, t);
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 fa[FN], fb[FM];
#pragma unroll
        for (int i = 0; i < FN; ++i) fa[i] = frag_bf16<A_KC, TILE_N>(As, wn * WN + i * 32 + l31, kk, hi);
#pragma unroll
        for (int j = 0; j < FM; ++j) fb[j] = frag_bf16<B_KC, TILE_M>(Bs, wm * WM + j * 32 + l31, kk, hi);
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
      for (int s = 0; s < BK / 2; ++s) {
        int k = 2 * s + hi;
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
    __syncthreads();
  }

  // ---- epilogue: lane owns output row n, columns m .. m+3 for each accumulator quad g ------------------------
  const smx_epilogue& e = p.e;
  const float* bias = e.bias ? e.bias + (long)bz * e.bias_batch_stride : nullptr;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = n0 + wn * WN + i * 32 + l31;
    if (n >= p.N) continue;
    float mk = 1.f;
    if (e.row_mask) mk = e.row_mask[n] ? 1.f : 0.f;
    const long c0r = (e.c0_mode != SMX_C0_NONE) ? c0_row(e, n) : 0;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = m0 + wm * WM + j * 32 + g * 8 + hi * 4;
        if (m >= p.M) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][g * 4 + q];
        if (e.out_mode == SMX_OUT_ATOMIC_F32) {
          float* Cf = reinterpret_cast<float*>(p.C) + (long)bz * p.sC + (long)n * p.ldc + m;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (m + q < p.M) atomicAdd(Cf + q, e.alpha * v[q]);
          continue;
        }
        const bool full = p.epi_vec && (m + 3 < p.M);
        if (bias) {
          if (full) {
            float4 b4 = *reinterpret_cast<const float4*>(bias + m);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) v[q] += bias[m + q];
          }
        }
        if (e.c0_mode != SMX_C0_NONE) {
          const float* c0p = e.c0 + c0r * e.ldc0 + m;
          if (full) {
            float4 c4 = *reinterpret_cast<const float4*>(c0p);
            v[0] += c4.x; v[1] += c4.y; v[2] += c4.z; v[3] += c4.w;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) v[q] += c0p[q];
          }
        }
        if (e.z) {
          T* zp = reinterpret_cast<T*>(e.z) + (long)bz * p.sC + (long)n * e.ldz + m;
          if (full) store4<T>(zp, v);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) zp[q] = from_f32<T>(v[q]);
          }
        }
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = e.alpha * (act_fwd(e.act, v[q]) * mk);
        if (e.res) {
          const T* rp = reinterpret_cast<const T*>(e.res) + (long)bz * p.sC + (long)n * e.ldr + m;
          if (full) {
            float r4[4];
            load4<T>(rp, r4);
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] += r4[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) y[q] += to_f32(rp[q]);
          }
        }
        if (e.out_mode == SMX_OUT_F32) {
          float* cp = reinterpret_cast<float*>(p.C) + (long)bz * p.sC + (long)n * p.ldc + m;
          if (full) *reinterpret_cast<float4*>(cp) = make_float4(y[0], y[1], y[2], y[3]);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) cp[q] = y[q];
          }
        } else {
          T* cp = reinterpret_cast<T*>(p.C) + (long)bz * p.sC + (long)n * p.ldc + m;
          if (full) store4<T>(cp, y);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (m + q < p.M) cp[q] = from_f32<T>(y[q]);
          }
        }
      }
    }
  }
}

// ---- host dispatch ------------------------------------------------------------------------------------------
template <typename T, bool A_KC, bool B_KC, int TN, int TM>
static int launch_tile(GemmParams& p, bool vec, hipStream_t s) {
  p.tiles_n = (p.N + TN - 1) / TN;
  p.tiles_m = (p.M + TM - 1) / TM;
  dim3 grid(p.tiles_n * p.tiles_m, p.batch * p.splits);
  if (vec) hipLaunchKernelGGL((gemm_kernel<T, A_KC, B_KC, TN, TM, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<T, A_KC, B_KC, TN, TM, false>), grid, dim3(256), 0, s, p);
  return check_launch("smx_gemm");
}

template <typename T, bool A_KC, bool B_KC>
static int launch_layout(GemmParams& p, bool vec, hipStream_t s) {
  // big tiles once they alone fill the chip (256 CUs x 2 resident blocks); otherwise 64x64 for more blocks
  long big = (long)((p.N + 127) / 128) * ((p.M + 127) / 128) * p.batch * p.splits;
  if (big >= 384) return launch_tile<T, A_KC, B_KC, 128, 128>(p, vec, s);
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

extern "C" int smx_gemm(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B,
                        int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K,
                        int batch, int splits, const smx_epilogue* epi, void* stream) {
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
    SMX_REQUIRE(p.e.out_mode == SMX_OUT_ATOMIC_F32, "smx_gemm: splits>1 needs SMX_OUT_ATOMIC_F32");
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
    SMX_REQUIRE(!p.e.bias && !p.e.c0 && !p.e.z && !p.e.res && !p.e.row_mask && p.e.act == SMX_ACT_NONE,
                "smx_gemm: atomic output takes no epilogue");
  SMX_REQUIRE(p.e.c0_mode == SMX_C0_NONE || (p.e.c0 && (p.e.c0_mode == SMX_C0_ROW || p.e.c0_div > 0)),
              "smx_gemm: bad C0 spec");
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  const int vpt = dtype == SMX_BF16 ? 8 : 4;
  // vector (16-byte) operand loads need aligned bases / strides and whole vectors along the contiguous dim
  bool a_kc = layout != SMX_GEMM_TN, b_kc = layout == SMX_GEMM_NT;
  bool vec = aligned16(A) && aligned16(B) && lda % vpt == 0 && ldb % vpt == 0 && (strideA * es) % 16 == 0 &&
             (strideB * es) % 16 == 0;
  vec = vec && (a_kc ? K % vpt == 0 : N % vpt == 0) && (b_kc ? K % vpt == 0 : M % vpt == 0);
  // 4-wide epilogue accesses
  auto ok4 = [&](const void* ptr, int64_t ld, size_t esz) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) % (4 * esz)) == 0 && ld % 4 == 0);
  };
  size_t cs = p.e.out_mode == SMX_OUT_T ? es : 4;
  p.epi_vec = ok4(C, ldc, cs) && ok4(p.e.z, p.e.ldz, es) && ok4(p.e.res, p.e.ldr, es) && ok4(p.e.bias, 4, 4) &&
              ok4(p.e.c0, p.e.ldc0, 4) && (strideC % 4 == 0) && (p.e.bias_batch_stride % 4 == 0);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SMX_BF16) return launch_dtype<bf16_t>(layout, p, vec, s);
  return launch_dtype<float>(layout, p, vec, s);
}

extern "C" int smx_linear_act_mask_fwd(int dtype, const void* X, int64_t ldx, const void* W, int64_t ldw, void* Y,
                                       int64_t ldy, int N, int M, int K, const smx_epilogue* epi, void* stream) {
  return smx_gemm(SMX_GEMM_NT, dtype, X, ldx, 0, W, ldw, 0, Y, ldy, 0, N, M, K, 1, 1, epi, stream);
}
