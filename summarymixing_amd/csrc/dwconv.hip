// dwconv.hip — fused GLU + depthwise Conv1d over time (Conformer conv module) and the gated reflect-padded
// variant (Branchformer CSGU), forward and backward, gfx950.
//
// Tile = 64 frames x 64 channels per workgroup (256 threads: lane = channel, 4 waves x 16 frames).  The GLU'd
// input tile with its +-(k-1)/2 halo is built ONCE in LDS ([frame][channel] fp32: consecutive lanes hit
// consecutive banks, conflict free), so P is read from HBM exactly once per tile (+ halo) and u is never
// materialised in HBM.  Dynamic Chunk Convolution only changes a per-tap predicate.
#include "smx_common.h"

namespace smx {

constexpr int DW_TT = 64;      // frames per tile
constexpr int DW_CT = 64;      // channels per tile
constexpr int DW_KMAX = 33;    // max taps
constexpr int DW_ROWS = DW_TT + DW_KMAX - 1;

struct DwParams {
  const void* P; long ldp;
  const float* w; const float* bias;
  const void* gate; long ldg;
  void* Y; long ldy;          // fwd out / bwd: dY in
  void* dP; long lddp;
  void* dgate; long lddg;
  float* dw; float* dbias;
  int B, T, D, k, glu, pad_mode, chunk;
  unsigned dthresh; float dscale; uint64_t dseed; const uint64_t* epoch;   // fused dropout of the output (rolling CSGU forward only)
};

__device__ __forceinline__ int map_frame(int tau, int T, int pad_mode) {
  if (tau >= 0 && tau < T) return tau;
  if (pad_mode == SMX_PAD_REFLECT) {
    int r = tau < 0 ? -tau : 2 * (T - 1) - tau;
    return (r >= 0 && r < T) ? r : -1;
  }
  return -1;
}

// fill the u tile (rows t0-pad .. t0+TT+pad) for channel c of batch b
template <typename T>
__device__ __forceinline__ void fill_u(float (*u)[DW_CT], const DwParams& p, int b, int t0, int pad, int ch, int cl, int wv) {
  const T* P = reinterpret_cast<const T*>(p.P);
  const int rows = DW_TT + 2 * pad;
  for (int i = wv; i < rows; i += 4) {
    int src = map_frame(t0 - pad + i, p.T, p.pad_mode);
    float val = 0.f;
    if (src >= 0 && ch < p.D) {
      const T* row = P + ((long)b * p.T + src) * p.ldp;
      float a = to_f32(row[ch]);
      val = p.glu ? a * sigmoidf_(to_f32(row[p.D + ch])) : a;
    }
    u[i][cl] = val;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwParams p) {
  __shared__ float u[DW_ROWS][DW_CT];
  __shared__ float wl[DW_KMAX][DW_CT];
  const int cl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ch = blockIdx.x * DW_CT + cl, t0 = blockIdx.y * DW_TT, b = blockIdx.z;
  const int pad = (p.k - 1) / 2;
  fill_u<T>(u, p, b, t0, pad, ch, cl, wv);
  for (int j = wv; j < p.k; j += 4) wl[j][cl] = ch < p.D ? p.w[(long)ch * p.k + j] : 0.f;
  __syncthreads();
  if (ch >= p.D) return;
  const float bs = p.bias ? p.bias[ch] : 0.f;
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = bs;
  const int f0 = wv * 16;
  if (p.chunk > 0) {
    for (int j = 0; j < p.k; ++j) {
      const float wj = wl[j][cl];
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        int t = t0 + f0 + o, tau = t + j - pad;
        int lim = (t / p.chunk + 1) * p.chunk;
        acc[o] += (tau < lim) ? wj * u[f0 + o + j][cl] : 0.f;
      }
    }
  } else {
    for (int j = 0; j < p.k; ++j) {
      const float wj = wl[j][cl];
#pragma unroll
      for (int o = 0; o < 16; ++o) acc[o] += wj * u[f0 + o + j][cl];
    }
  }
  T* Y = reinterpret_cast<T*>(p.Y);
  const T* G = reinterpret_cast<const T*>(p.gate);
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    int t = t0 + f0 + o;
    if (t < p.T) {
      float y = acc[o];
      if (G) y *= to_f32(G[((long)b * p.T + t) * p.ldg + ch]);
      Y[((long)b * p.T + t) * p.ldy + ch] = from_f32<T>(y);
    }
  }
}

// backward: persistent blocks over (b, t-tile) pairs of one channel tile; dw accumulates in LDS.
// Tap / bias gradients: every wave keeps its own sums in registers over all its tiles, the four waves are folded in a fixed
// order at the end and the workgroup writes ONE partial row [D][k + 1] (partial != NULL: reduced by dw_partials_reduce_kernel,
// bit-reproducible - round 3; LDS and global fp32 atomics before) or, without a workspace, adds it to dw / dbias atomically.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwParams p, int tiles_t, float* __restrict__ partial) {
  __shared__ float u[DW_ROWS][DW_CT];
  __shared__ float g[DW_ROWS][DW_CT];     // dYg with halo
  __shared__ float wl[DW_KMAX][DW_CT];
  __shared__ float red[3][DW_CT];
  const int cl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ch = blockIdx.x * DW_CT + cl;
  const int pad = (p.k - 1) / 2;
  const int rows = DW_TT + 2 * pad;
  for (int j = wv; j < p.k; j += 4) wl[j][cl] = ch < p.D ? p.w[(long)ch * p.k + j] : 0.f;
  float dwr[DW_KMAX], dbr = 0.f;
#pragma unroll
  for (int j = 0; j < DW_KMAX; ++j) dwr[j] = 0.f;
  const T* dY = reinterpret_cast<const T*>(p.Y);
  const T* G = reinterpret_cast<const T*>(p.gate);
  const T* P = reinterpret_cast<const T*>(p.P);
  T* dP = reinterpret_cast<T*>(p.dP);
  T* dG = reinterpret_cast<T*>(p.dgate);
  const float bs = (p.bias && ch < p.D) ? p.bias[ch] : 0.f;
  const int f0 = wv * 16;
  const long total = (long)p.B * tiles_t;
  for (long it = blockIdx.y; it < total; it += gridDim.y) {
    const int b = (int)(it / tiles_t), t0 = (int)(it % tiles_t) * DW_TT;
    __syncthreads();
    fill_u<T>(u, p, b, t0, pad, ch, cl, wv);
    for (int i = wv; i < rows; i += 4) {
      int t = t0 - pad + i;
      float val = 0.f;
      if (t >= 0 && t < p.T && ch < p.D) {
        val = to_f32(dY[((long)b * p.T + t) * p.ldy + ch]);
        if (G) val *= to_f32(G[((long)b * p.T + t) * p.ldg + ch]);
      }
      g[i][cl] = val;
    }
    __syncthreads();
    if (ch < p.D) {
      // (1) dgate = dY * conv(t)   (forward recompute), dbias, dw partials
      float dbp = 0.f;
      if (dG) {
        float acc[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = bs;
        for (int j = 0; j < p.k; ++j) {
          const float wj = wl[j][cl];
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            int t = t0 + f0 + o, tau = t + j - pad;
            bool ok = p.chunk > 0 ? (tau < (t / p.chunk + 1) * p.chunk) : true;
            acc[o] += ok ? wj * u[f0 + o + j][cl] : 0.f;
          }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          int t = t0 + f0 + o;
          if (t < p.T) dG[((long)b * p.T + t) * p.lddg + ch] = from_f32<T>(to_f32(dY[((long)b * p.T + t) * p.ldy + ch]) * acc[o]);
        }
      }
#pragma unroll
      for (int j = 0; j < DW_KMAX; ++j) {
        if (j < p.k) {                                    // (uniform)
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            int t = t0 + f0 + o, tau = t + j - pad;
            bool ok = p.chunk > 0 ? (tau < (t / p.chunk + 1) * p.chunk) : true;
            s += ok ? g[pad + f0 + o][cl] * u[f0 + o + j][cl] : 0.f;
          }
          dwr[j] += s;
        }
      }
#pragma unroll
      for (int o = 0; o < 16; ++o) dbp += g[pad + f0 + o][cl];
      dbr += dbp;
      // (2) du(tau) = sum_j w_j dYg(tau - j + pad)  [+ reflect folds], then GLU backward
      float du[16];
#pragma unroll
      for (int o = 0; o < 16; ++o) du[o] = 0.f;
      for (int j = 0; j < p.k; ++j) {
        const float wj = wl[j][cl];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          int tau = t0 + f0 + o, t = tau - j + pad;       // row of g: (t - (t0 - pad)) = f0 + o - j + 2 pad
          bool ok = p.chunk > 0 ? (t >= 0 && tau < (t / p.chunk + 1) * p.chunk) : true;
          du[o] += ok ? wj * g[f0 + o - j + 2 * pad][cl] : 0.f;
        }
      }
      if (p.pad_mode == SMX_PAD_REFLECT) {
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          int tau = t0 + f0 + o;
          if (tau >= 1 && tau <= pad) {              // virtual frame v = -tau reads frame tau
            int v = -tau;
            float s = 0.f;
            for (int j = 0; j < p.k; ++j) {
              int t = v - j + pad, r = t - (t0 - pad);
              if (t >= 0 && t < p.T && r >= 0 && r < rows) s += wl[j][cl] * g[r][cl];
            }
            du[o] += s;
          }
          if (tau <= p.T - 2 && tau >= p.T - 1 - pad) {   // virtual frame v = 2(T-1)-tau
            int v = 2 * (p.T - 1) - tau;
            float s = 0.f;
            for (int j = 0; j < p.k; ++j) {
              int t = v - j + pad, r = t - (t0 - pad);
              if (t >= 0 && t < p.T && r >= 0 && r < rows) s += wl[j][cl] * g[r][cl];
            }
            du[o] += s;
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        int t = t0 + f0 + o;
        if (t < p.T) {
          const T* row = P + ((long)b * p.T + t) * p.ldp;
          T* drow = dP + ((long)b * p.T + t) * p.lddp;
          if (p.glu) {
            float a = to_f32(row[ch]), s = sigmoidf_(to_f32(row[p.D + ch]));
            drow[ch] = from_f32<T>(du[o] * s);
            drow[p.D + ch] = from_f32<T>(du[o] * a * s * (1.f - s));
          } else {
            drow[ch] = from_f32<T>(du[o]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j <= DW_KMAX; ++j) {
    if (j < p.k || j == DW_KMAX) {                        // (uniform; j == DW_KMAX: the bias gradient)
      const float v = j < DW_KMAX ? dwr[j < DW_KMAX ? j : 0] : dbr;
      __syncthreads();
      if (wv > 0) red[wv - 1][cl] = v;
      __syncthreads();
      if (wv == 0 && ch < p.D) {
        const float tot = ((v + red[0][cl]) + red[1][cl]) + red[2][cl];
        const int jj = j < DW_KMAX ? j : p.k;
        if (partial) partial[((long)blockIdx.y * p.D + ch) * (p.k + 1) + jj] = tot;
        else if (j < DW_KMAX) atomicAdd(p.dw + (long)ch * p.k + j, tot);
        else if (p.dbias) atomicAdd(p.dbias + ch, tot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Register-window specialisations for the production shape (compile-time K taps, zero padding, no chunking,
// no gate, D % 8 == 0): 16-byte global accesses everywhere (a tile row of 64 channels is one 128 B line for bf16),
// each thread pulls its 16+K-1 input frames from LDS ONCE and does all K x 16 FMAs out of registers.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct DwVec { static constexpr int N = 16 / sizeof(T); };

template <typename T>
__device__ __forceinline__ void ld_chunk(const T* p, float (&f)[DwVec<T>::N]) {
  if constexpr (sizeof(T) == 2) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_bits_to_f32(w[i] & 0xffffu); f[2 * i + 1] = bf16_bits_to_f32(w[i] >> 16); }
  } else {
    const float4 r = *reinterpret_cast<const float4*>(p);
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
  }
}
template <typename T>
__device__ __forceinline__ void st_chunk(T* p, const float (&f)[DwVec<T>::N]) {
  if constexpr (sizeof(T) == 2) {
    uint4 r;
    r.x = pack_bf16x2(f[0], f[1]); r.y = pack_bf16x2(f[2], f[3]);
    r.z = pack_bf16x2(f[4], f[5]); r.w = pack_bf16x2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = r;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
}

// u tile rows [t0-PAD, t0+TT+PAD) x 64 channels, GLU applied, zero padded in time; 16 B per thread per access
template <typename T, int ROWS, int PAD>
__device__ __forceinline__ void fill_u_vec(float (*u)[DW_CT], const DwParams& p, int b, int t0, int c0, int tid) {
  constexpr int VW = DwVec<T>::N, LPR = DW_CT / VW, NITEM = (ROWS * LPR + 255) / 256;
  const T* P = reinterpret_cast<const T*>(p.P);
#pragma unroll
  for (int k = 0; k < NITEM; ++k) {
    const int it = tid + 256 * k;
    if (it < ROWS * LPR) {
      const int i = it / LPR, cc = (it % LPR) * VW;
      const int tau = map_frame(t0 - PAD + i, p.T, p.pad_mode), ch = c0 + cc;   // zero pad -> -1, reflect -> mirrored
      float v[VW];
#pragma unroll
      for (int q = 0; q < VW; ++q) v[q] = 0.f;
      if (tau >= 0 && ch < p.D) {
        const T* row = P + ((long)b * p.T + tau) * p.ldp;
        ld_chunk<T>(row + ch, v);
        if (p.glu) {
          float gte[VW];
          ld_chunk<T>(row + p.D + ch, gte);
#pragma unroll
          for (int q = 0; q < VW; ++q) v[q] *= sigmoidf_(gte[q]);
        }
      }
#pragma unroll
      for (int q4 = 0; q4 < VW / 4; ++q4)
        *reinterpret_cast<float4*>(&u[i][cc + 4 * q4]) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
    }
  }
}

template <typename T, int K, bool GATE>
__global__ __launch_bounds__(256) void dwconv_fwd_fast(DwParams p) {
  constexpr int PAD = (K - 1) / 2, WIN = 16 + K - 1, ROWS = DW_TT + K - 1;
  constexpr int VW = DwVec<T>::N, LPR = DW_CT / VW;
  __shared__ __attribute__((aligned(16))) float u[ROWS][DW_CT];
  const int cl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c0 = blockIdx.x * DW_CT, ch = c0 + cl, t0 = blockIdx.y * DW_TT, b = blockIdx.z;
  fill_u_vec<T, ROWS, PAD>(u, p, b, t0, c0, threadIdx.x);
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = ch < p.D ? p.w[(long)ch * K + j] : 0.f;
  const float bs = (p.bias && ch < p.D) ? p.bias[ch] : 0.f;
  __syncthreads();
  const int f0 = wv * 16;
  float win[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = u[f0 + i][cl];
  __syncthreads();                                   // every window is in registers: the tile can be reused
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float acc = bs;
#pragma unroll
    for (int j = 0; j < K; ++j) acc += w[j] * win[o + j];
    u[f0 + o][cl] = acc;
  }
  __syncthreads();
  T* Y = reinterpret_cast<T*>(p.Y);
  for (int it = threadIdx.x; it < DW_TT * LPR; it += 256) {
    const int r = it / LPR, cc = (it % LPR) * VW;
    const int t = t0 + r, chv = c0 + cc;
    if (t < p.T && chv < p.D) {
      float v[VW];
#pragma unroll
      for (int q = 0; q < VW; ++q) v[q] = u[r][cc + q];
      if constexpr (GATE) {                            // CSGU: y = conv(x2) * x1
        float gt[VW];
        ld_chunk<T>(reinterpret_cast<const T*>(p.gate) + ((long)b * p.T + t) * p.ldg + chv, gt);
#pragma unroll
        for (int q = 0; q < VW; ++q) v[q] *= gt[q];
      }
      st_chunk<T>(Y + ((long)b * p.T + t) * p.ldy + chv, v);
    }
  }
}

template <typename T, int K, bool REFLECT, bool GATE>
__global__ __launch_bounds__(256) void dwconv_bwd_fast(DwParams p, int tiles_t, float* __restrict__ partial, int gy) {
  constexpr int PAD = (K - 1) / 2, WIN = 16 + K - 1, ROWS = DW_TT + K - 1;
  constexpr int VW = DwVec<T>::N, LPR = DW_CT / VW, NITEM = (ROWS * LPR + 255) / 256;
  __shared__ __attribute__((aligned(16))) float u[ROWS][DW_CT];
  __shared__ __attribute__((aligned(16))) float g[ROWS][DW_CT];
  // CSGU: the tile's own dY rows, kept for dgate = dY * conv (re-reading them from memory after the conv recompute was
  // one more exposed round trip per tile)
  __shared__ __attribute__((aligned(16))) float dyc[GATE ? DW_TT : 1][DW_CT];
  __shared__ float red[3][DW_CT];
  __shared__ float wl[REFLECT ? K : 1][DW_CT];         // the taps once more, indexable at run time by the edge fold
  const int cl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // workgroup b runs on XCD b % 8: keep the channel tiles of one frame-tile sequence on ONE XCD and adjacent in
  // dispatch order, so the 128-byte pieces of a feature row are fetched by neighbours at the same time
  const int ctiles = (p.D + DW_CT - 1) / DW_CT;
  const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
  const int by = (widx / ctiles) * 8 + xcd, bx = widx % ctiles;
  if (by >= gy) return;
  const int c0 = bx * DW_CT, ch = c0 + cl;
  const bool cok = ch < p.D;
  float w[K], dw[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { w[j] = cok ? p.w[(long)ch * K + j] : 0.f; dw[j] = 0.f; }
  if constexpr (REFLECT) {
    for (int j = wv; j < K; j += 4) wl[j][cl] = cok ? p.w[(long)ch * K + j] : 0.f;   // (first use is behind barriers)
  }
  float dbs = 0.f;
  const T* dY = reinterpret_cast<const T*>(p.Y);
  const T* P = reinterpret_cast<const T*>(p.P);
  T* dP = reinterpret_cast<T*>(p.dP);
  const int f0 = wv * 16;
  const long total = (long)p.B * tiles_t;
  for (long it0 = by; it0 < total; it0 += gy) {
    const int b = (int)(it0 / tiles_t), t0 = (int)(it0 % tiles_t) * DW_TT;
    __syncthreads();
    fill_u_vec<T, ROWS, PAD>(u, p, b, t0, c0, threadIdx.x);
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {                // dY tile with halo, 16 B per thread
      const int it = threadIdx.x + 256 * k;
      if (it < ROWS * LPR) {
        const int i = it / LPR, cc = (it % LPR) * VW;
        const int t = t0 - PAD + i, chv = c0 + cc;
        float v[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q) v[q] = 0.f;
        if (t >= 0 && t < p.T && chv < p.D) {
          ld_chunk<T>(dY + ((long)b * p.T + t) * p.ldy + chv, v);
          if constexpr (GATE) {                        // gradient w.r.t. the conv output: dY * gate
            float gt[VW];
            ld_chunk<T>(reinterpret_cast<const T*>(p.gate) + ((long)b * p.T + t) * p.ldg + chv, gt);
            if (i >= PAD && i < PAD + DW_TT) {
#pragma unroll
              for (int q4 = 0; q4 < VW / 4; ++q4)
                *reinterpret_cast<float4*>(&dyc[i - PAD][cc + 4 * q4]) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
            }
#pragma unroll
            for (int q = 0; q < VW; ++q) v[q] *= gt[q];
          }
        }
#pragma unroll
        for (int q4 = 0; q4 < VW / 4; ++q4)
          *reinterpret_cast<float4*>(&g[i][cc + 4 * q4]) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
      }
    }
    __syncthreads();
    float uw[WIN], gw[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { uw[i] = u[f0 + i][cl]; gw[i] = g[f0 + i][cl]; }
    __syncthreads();                                 // windows are in registers: u can take the du tile
    if constexpr (GATE) {                            // dgate = dY * conv(x2)  (forward recompute out of the window)
      const float bsv = (p.bias && cok) ? p.bias[ch] : 0.f;
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        float acc = bsv;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += w[j] * uw[o + j];
        u[f0 + o][cl] = acc;
      }
      __syncthreads();
      T* dG = reinterpret_cast<T*>(p.dgate);
      for (int it = threadIdx.x; it < DW_TT * LPR; it += 256) {
        const int r = it / LPR, cc = (it % LPR) * VW;
        const int t = t0 + r, chv = c0 + cc;
        if (t < p.T && chv < p.D) {
          float o8[VW];
#pragma unroll
          for (int q = 0; q < VW; ++q) o8[q] = dyc[r][cc + q] * u[r][cc + q];
          st_chunk<T>(dG + ((long)b * p.T + t) * p.lddg + chv, o8);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      float sacc = 0.f;
#pragma unroll
      for (int o = 0; o < 16; ++o) sacc += gw[PAD + o] * uw[o + j];
      dw[j] += sacc;
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) dbs += gw[PAD + o];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      float du = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) du += w[j] * gw[o - j + 2 * PAD];
      u[f0 + o][cl] = du;
    }
    if constexpr (REFLECT) {
      // fold the gradient of the mirrored virtual frames into the edge frames.  Kept OUT of the unrolled FMA block above
      // (a rolled loop over this thread's own 16 rows of the du tile, entered only by tiles that touch an utterance edge)
      // and with the taps read from an LDS copy: branches inside the unrolled block and a run-time index into the
      // register array w[] made every tile of the reflect variant 3.3x slower; taps re-read from global memory cost a
      // dependent L2 round trip per tap (T = 250: half the tiles touch an edge, 451 us vs 322 us for zero padding).
      if (t0 <= PAD || t0 + DW_TT + PAD >= p.T - 1) {
        __syncthreads();                               // every du row of the tile is in LDS
        // the PAD rows next to either edge are shared among the 4 waves.  The wave index is made explicitly uniform: row,
        // tap range and LDS row addresses are then scalar arithmetic, the tap loops are plain counted loops (no per-lane
        // predication) and unroll by 4 so that their LDS reads overlap
        const int wvs = __builtin_amdgcn_readfirstlane(wv);
        for (int idx = wvs; idx < 2 * PAD; idx += 4) {
          const bool top = idx < PAD;
          const int tau = top ? 1 + idx : p.T - 1 - PAD + (idx - PAD);
          const int o = tau - t0;
          // frames 1..PAD are mirrored at the top, frames T-1-PAD..T-2 at the bottom; for T < 32 a frame can sit in both
          // ranges: its top pass then does both folds and the bottom pass skips it
          const bool in_top = tau >= 1 && tau <= PAD && tau <= p.T - 1;
          const bool in_bot = tau >= 0 && tau <= p.T - 2 && tau >= p.T - 1 - PAD;
          if (o < 0 || o >= DW_TT || (top ? !in_top : (!in_bot || in_top))) continue;
          float add = 0.f;
          if (in_top) {                                // virtual frame -tau mirrors frame tau: taps j <= PAD - tau
            const int r0 = 2 * PAD - tau - t0;         // g row of tap 0; tap j reads row r0 - j (frame PAD - tau - j)
            const int jlo = max(max(0, r0 - (ROWS - 1)), PAD - tau - p.T + 1), jhi = min(PAD - tau, r0);
#pragma unroll 4
            for (int j = jlo; j <= jhi; ++j) add += wl[j][cl] * g[r0 - j][cl];
          }
          if (in_bot) {                                // virtual frame 2(T-1)-tau mirrors frame tau: taps j >= T-1-tau+PAD
            const int r0 = 2 * (p.T - 1) - tau + 2 * PAD - t0;   // tap j reads row r0 - j (frame 2(T-1) - tau - j + PAD)
            const int jlo = max(max(0, p.T - 1 - tau + PAD), r0 - (ROWS - 1));
            const int jhi = min(min(K - 1, r0), 2 * (p.T - 1) - tau + PAD);
#pragma unroll 4
            for (int j = jlo; j <= jhi; ++j) add += wl[j][cl] * g[r0 - j][cl];
          }
          u[o][cl] += add;
        }
      }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < DW_TT * LPR; it += 256) {   // GLU backward, 16 B loads / stores
      const int r = it / LPR, cc = (it % LPR) * VW;
      const int t = t0 + r, chv = c0 + cc;
      if (t < p.T && chv < p.D) {
        float du[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q) du[q] = u[r][cc + q];
        T* drow = dP + ((long)b * p.T + t) * p.lddp;
        if (p.glu) {
          const T* row = P + ((long)b * p.T + t) * p.ldp;
          float a[VW], gt[VW], d1[VW], d2[VW];
          ld_chunk<T>(row + chv, a);
          ld_chunk<T>(row + p.D + chv, gt);
#pragma unroll
          for (int q = 0; q < VW; ++q) {
            const float sg = sigmoidf_(gt[q]);
            d1[q] = du[q] * sg;
            d2[q] = du[q] * a[q] * sg * (1.f - sg);
          }
          st_chunk<T>(drow + chv, d1);
          st_chunk<T>(drow + p.D + chv, d2);
        } else {
          st_chunk<T>(drow + chv, du);
        }
      }
    }
  }
  // flush dw / dbias: 4 waves -> LDS -> one partial row per block (reduced in a fixed order by
  // dw_partials_reduce_kernel; same-address fp32 atomics from ~1000 blocks cost ~100 ns each)
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    const float v = j < K ? dw[j < K ? j : 0] : dbs;
    __syncthreads();
    if (wv > 0) red[wv - 1][cl] = v;
    __syncthreads();
    if (wv == 0 && cok) {
      const float tot = ((v + red[0][cl]) + red[1][cl]) + red[2][cl];
      partial[((long)by * p.D + ch) * (K + 1) + j] = tot;
    }
  }
}

// dw[ch][j] += sum_y partial[y][ch][j] (j < K); dbias[ch] += sum_y partial[y][ch][K]
__global__ __launch_bounds__(256) void dw_partials_reduce_kernel(const float* __restrict__ partial, int ny, int D, int K,
                                                                 float* dw, float* dbias) {
  const long W = (long)D * (K + 1);
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= W) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int y = 0;
  for (; y + 3 < ny; y += 4) {
    s0 += partial[(long)y * W + idx]; s1 += partial[(long)(y + 1) * W + idx];
    s2 += partial[(long)(y + 2) * W + idx]; s3 += partial[(long)(y + 3) * W + idx];
  }
  for (; y < ny; ++y) s0 += partial[(long)y * W + idx];
  const int ch = (int)(idx / (K + 1)), j = (int)(idx % (K + 1));
  const float tot = (s0 + s1) + (s2 + s3);
  if (j < K) dw[(long)ch * K + j] += tot;
  else if (dbias) dbias[ch] += tot;
}

}  // namespace smx

#include "dwconv_roll.h"

using namespace smx;

// does a rolling register-window path (dwconv_roll.h) take this call?  1 = GLU + zero padding (bf16 and fp32),
// 2 = CSGU gate + reflect padding (bf16)
static int roll_kind(int dtype, int T, int D, int k, int glu, int pad_mode, int chunk, bool has_gate) {
  if (k != 31 || D % 64 != 0) return 0;
  // (Dynamic Chunk Convolution, chunk > 0, is part of the GLU kernels)
  if (glu && !has_gate && pad_mode == SMX_PAD_ZERO) return 1;
  if (chunk <= 0 && !glu && has_gate && pad_mode == SMX_PAD_REFLECT && dtype == SMX_BF16 && T > 15) return 2;
  return 0;
}

static int dwconv_fwd_impl(int dtype, const void* P, int64_t ldp, const float* w, const float* bias, const void* gate, int64_t ldg,
                           void* Y, int64_t ldy, int B, int T, int D, int k, int glu, int pad_mode, int chunk, float drop_p,
                           uint64_t drop_seed, const uint64_t* epoch, void* stream);

extern "C" int smx_dwconv1d_glu_fwd(int dtype, const void* P, int64_t ldp, const float* w, const float* bias,
                                    const void* gate, int64_t ldg, void* Y, int64_t ldy, int B, int T, int D, int k,
                                    int glu, int pad_mode, int chunk, void* stream) {
  return dwconv_fwd_impl(dtype, P, ldp, w, bias, gate, ldg, Y, ldy, B, T, D, k, glu, pad_mode, chunk, 0.f, 0, nullptr, stream);
}

extern "C" int smx_dwconv1d_glu_fwd_drop(int dtype, const void* P, int64_t ldp, const float* w, const float* bias,
                                         const void* gate, int64_t ldg, void* Y, int64_t ldy, int B, int T, int D, int k,
                                         int glu, int pad_mode, int chunk, float drop_p, uint64_t drop_seed, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "smx_dwconv1d_glu_fwd_drop: 0 <= drop_p < 1");
  return dwconv_fwd_impl(dtype, P, ldp, w, bias, gate, ldg, Y, ldy, B, T, D, k, glu, pad_mode, chunk, drop_p, drop_seed, epoch, stream);
}

static int dwconv_fwd_impl(int dtype, const void* P, int64_t ldp, const float* w, const float* bias, const void* gate, int64_t ldg,
                           void* Y, int64_t ldy, int B, int T, int D, int k, int glu, int pad_mode, int chunk, float drop_p,
                           uint64_t drop_seed, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(P && w && Y, "smx_dwconv1d_glu_fwd: null pointer");
  SMX_REQUIRE(k >= 1 && k <= DW_KMAX && (k & 1), "smx_dwconv1d_glu_fwd: k=%d must be odd and <= %d", k, DW_KMAX);
  SMX_REQUIRE(pad_mode != SMX_PAD_REFLECT || (k - 1) / 2 < T, "smx_dwconv1d_glu_fwd: reflect pad needs (k-1)/2 < T");
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  DwParams p;
  memset(&p, 0, sizeof(p));
  p.P = P; p.ldp = ldp; p.w = w; p.bias = bias; p.gate = gate; p.ldg = ldg; p.Y = Y; p.ldy = ldy;
  p.B = B; p.T = T; p.D = D; p.k = k; p.glu = glu; p.pad_mode = pad_mode; p.chunk = chunk;
  dim3 grid((D + DW_CT - 1) / DW_CT, (T + DW_TT - 1) / DW_TT, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vw = dtype == SMX_BF16 ? 8 : 4;
  const bool fast = k == 31 && chunk <= 0 && D % vw == 0 && ldp % vw == 0 && ldy % vw == 0 && aligned16(P) && aligned16(Y) &&
                    (gate == nullptr || (ldg % vw == 0 && aligned16(gate))) && (pad_mode == SMX_PAD_ZERO || T > 15);
  int rk = roll_kind(dtype, T, D, k, glu, pad_mode, chunk, gate != nullptr);
  // the bf16 kernels move 16-byte pieces (LDS-DMA in, dwordx4 out): rows and bases must be 16-byte aligned, else tiled
  if (rk && dtype == SMX_BF16 && !(ldp % 8 == 0 && ldy % 8 == 0 && aligned16(P) && aligned16(Y) &&
                                   (gate == nullptr || (ldg % 8 == 0 && aligned16(gate))))) rk = 0;
  if (rk) {
    const long ldmax = ldp > ldy ? (ldp > ldg ? ldp : ldg) : (ldy > ldg ? ldy : ldg);
    SMX_REQUIRE((long)T * ldmax * 4 < (1L << 31), "smx_dwconv1d_glu_fwd: utterance span >= 2 GB (SMX_DWROLL=0 selects the tiled kernel)");
    int seg, nseg, gy;
    roll_geometry(B, T, D, &seg, &nseg, &gy);
    dim3 g1((unsigned)(8 * (D / 64) * ((gy + 7) / 8)));
    if (drop_p > 0.f && rk != 2) return fail(SMX_EUNSUPPORTED, "smx_dwconv1d_glu_fwd_drop: fused output dropout needs the rolling CSGU kernel");
    if (rk == 2) {
      p.dthresh = (unsigned)((double)drop_p * 4294967296.0); p.dscale = 1.f / (1.f - drop_p); p.dseed = drop_seed; p.epoch = epoch;
      hipLaunchKernelGGL(dwconv_rollc_fwd, g1, dim3(256), 0, s, p, seg, nseg, gy);
    }
    else if (dtype == SMX_BF16) {
      if (chunk > 0) hipLaunchKernelGGL((dwconv_rolls_fwd<true>), g1, dim3(256), 0, s, p, seg, nseg, gy);
      else hipLaunchKernelGGL((dwconv_rolls_fwd<false>), g1, dim3(256), 0, s, p, seg, nseg, gy);
    } else if (chunk > 0) hipLaunchKernelGGL((dwconv_roll_fwd<float, true>), g1, dim3(256), 0, s, p, seg, nseg, gy);
    else hipLaunchKernelGGL((dwconv_roll_fwd<float, false>), g1, dim3(256), 0, s, p, seg, nseg, gy);
  } else if (drop_p > 0.f) {
    return fail(SMX_EUNSUPPORTED, "smx_dwconv1d_glu_fwd_drop: fused output dropout needs the rolling CSGU kernel");
  } else if (fast) {
    if (gate) {
      if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_fwd_fast<bf16_t, 31, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((dwconv_fwd_fast<float, 31, true>), grid, dim3(256), 0, s, p);
    } else {
      if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_fwd_fast<bf16_t, 31, false>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((dwconv_fwd_fast<float, 31, false>), grid, dim3(256), 0, s, p);
    }
  } else if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((dwconv_fwd_kernel<float>), grid, dim3(256), 0, s, p);
  return check_launch("smx_dwconv1d_glu_fwd");
}

static long tiled_rows(int B, int T, int D) {
  const int tiles_t = (T + DW_TT - 1) / DW_TT, ctiles = (D + DW_CT - 1) / DW_CT;
  long total = (long)B * tiles_t, gy = (1024 + ctiles - 1) / ctiles;
  if (gy > total) gy = total;
  if (gy < 1) gy = 1;
  return gy;
}

extern "C" int smx_dwconv1d_glu_bwd_partial_rows(int dtype, int B, int T, int D, int k, int glu, int pad_mode, int chunk,
                                                 int has_gate) {
  if (B <= 0 || T <= 0 || D <= 0) return 0;
  if (roll_kind(dtype, T, D, k, glu, pad_mode, chunk, has_gate != 0)) {
    int seg, nseg, gy;
    roll_geometry(B, T, D, &seg, &nseg, &gy);
    return gy;
  }
  return (int)tiled_rows(B, T, D);
}

extern "C" size_t smx_dwconv1d_glu_bwd_workspace(int B, int T, int D, int k) {
  if (B <= 0 || T <= 0 || D <= 0) return 0;
  long gy = tiled_rows(B, T, D);
  if (D % 64 == 0 && k == 31) {
    int seg, nseg, gr;
    roll_geometry(B, T, D, &seg, &nseg, &gr);
    if (gr > gy) gy = gr;
  }
  return (size_t)gy * D * (k + 1) * sizeof(float);
}

extern "C" int smx_dwconv1d_glu_bwd(int dtype, const void* dY, int64_t lddy, const void* P, int64_t ldp, const float* w,
                                    const float* bias, const void* gate, int64_t ldg, void* dP, int64_t lddp,
                                    void* dgate, int64_t lddg, float* dw, float* dbias, int B, int T, int D, int k,
                                    int glu, int pad_mode, int chunk, void* workspace, void* stream) {
  SMX_REQUIRE(dY && P && w && dP && (dw || workspace), "smx_dwconv1d_glu_bwd: null pointer");
  SMX_REQUIRE(k >= 1 && k <= DW_KMAX && (k & 1), "smx_dwconv1d_glu_bwd: k=%d must be odd and <= %d", k, DW_KMAX);
  SMX_REQUIRE((gate == nullptr) == (dgate == nullptr), "smx_dwconv1d_glu_bwd: gate and dgate go together");
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  DwParams p;
  memset(&p, 0, sizeof(p));
  p.P = P; p.ldp = ldp; p.w = w; p.bias = bias; p.gate = gate; p.ldg = ldg; p.Y = const_cast<void*>(dY); p.ldy = lddy;
  p.dP = dP; p.lddp = lddp; p.dgate = dgate; p.lddg = lddg; p.dw = dw; p.dbias = dbias;
  p.B = B; p.T = T; p.D = D; p.k = k; p.glu = glu; p.pad_mode = pad_mode; p.chunk = chunk;
  const int tiles_t = (T + DW_TT - 1) / DW_TT;
  const int ctiles = (D + DW_CT - 1) / DW_CT;
  long total = (long)B * tiles_t;
  long gy = (1024 + ctiles - 1) / ctiles;          // ~1024 persistent blocks in total
  if (gy > total) gy = total;
  if (gy < 1) gy = 1;
  dim3 grid(ctiles, (unsigned)gy);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vw = dtype == SMX_BF16 ? 8 : 4;
  const bool fast = k == 31 && chunk <= 0 && D % vw == 0 && ldp % vw == 0 && lddy % vw == 0 && lddp % vw == 0 &&
                    aligned16(P) && aligned16(dY) && aligned16(dP) && workspace != nullptr &&
                    (gate == nullptr || (ldg % vw == 0 && lddg % vw == 0 && aligned16(gate) && aligned16(dgate))) &&
                    (pad_mode == SMX_PAD_ZERO || T > 15);
  int rk = workspace ? roll_kind(dtype, T, D, k, glu, pad_mode, chunk, gate != nullptr) : 0;
  if (rk && dtype == SMX_BF16 && !(ldp % 8 == 0 && lddy % 8 == 0 && lddp % 8 == 0 && aligned16(P) && aligned16(dY) && aligned16(dP) &&
                                   (gate == nullptr || (ldg % 8 == 0 && lddg % 8 == 0 && aligned16(gate) && aligned16(dgate))))) {
    // misaligned bf16 operands drop to the tiled kernel, which writes tiled_rows() partial rows - not the rolling count that
    // smx_dwconv1d_glu_bwd_partial_rows (it sees no pointers) reports for these sizes: a deferred reduction would fold the
    // wrong number of rows.  Deferred mode therefore refuses; the caller reduces immediately.
    if (!dw) return fail(SMX_EUNSUPPORTED, "smx_dwconv1d_glu_bwd: deferred reduction needs 16-byte aligned bf16 rows on the rolling path");
    rk = 0;
  }
  if (rk) {
    long ldmax = ldp > lddp ? (ldp > lddy ? ldp : lddy) : (lddp > lddy ? lddp : lddy);
    if (ldg > ldmax) ldmax = ldg;
    if (lddg > ldmax) ldmax = lddg;
    SMX_REQUIRE((long)T * ldmax * 4 < (1L << 31), "smx_dwconv1d_glu_bwd: utterance span >= 2 GB (SMX_DWROLL=0 selects the tiled kernel)");
    int seg, nseg, gr;
    roll_geometry(B, T, D, &seg, &nseg, &gr);
    float* partial = reinterpret_cast<float*>(workspace);
    dim3 g1((unsigned)(8 * (D / 64) * ((gr + 7) / 8)));
#ifdef SMX_DIAG
    const int abl = cfg().dwroll_ablate;
#endif
    if (rk == 2) {
      hipLaunchKernelGGL(dwconv_rollc_bwd, g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
      // the gradient of the mirrored virtual frames goes back to frames 1..15 / T-16..T-2 (30 rows per utterance)
      hipLaunchKernelGGL(dwconv_csgu_fold_kernel, dim3((unsigned)((D / 2 + 63) / 64), (unsigned)B), dim3(64), 0, s, p);
    } else if (dtype != SMX_BF16) {
      if (chunk > 0) hipLaunchKernelGGL((dwconv_roll_bwd<float, true>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
      else hipLaunchKernelGGL((dwconv_roll_bwd<float, false>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
    } else if (chunk > 0) hipLaunchKernelGGL((dwconv_rolls_bwd<0, true>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
#ifdef SMX_DIAG
    else if (abl == 1) hipLaunchKernelGGL((dwconv_rolls_bwd<1>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
    else if (abl == 2) hipLaunchKernelGGL((dwconv_rolls_bwd<2>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
#endif
    else hipLaunchKernelGGL((dwconv_rolls_bwd<0>), g1, dim3(256), 0, s, p, seg, nseg, gr, partial);
    const long W = (long)D * (k + 1);
    if (dw) hipLaunchKernelGGL(dw_partials_reduce_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, partial, gr, D, k, dw, dbias);
  } else if (fast) {
    float* partial = reinterpret_cast<float*>(workspace);
    dim3 g1((unsigned)(8 * ctiles * ((gy + 7) / 8)));
    const bool refl = pad_mode == SMX_PAD_REFLECT, gt = gate != nullptr;
#define DW_BWD(TT, R, G) hipLaunchKernelGGL((dwconv_bwd_fast<TT, 31, R, G>), g1, dim3(256), 0, s, p, tiles_t, partial, (int)gy)
    if (dtype == SMX_BF16) {
      if (refl && gt) DW_BWD(bf16_t, true, true); else if (refl) DW_BWD(bf16_t, true, false);
      else if (gt) DW_BWD(bf16_t, false, true); else DW_BWD(bf16_t, false, false);
    } else {
      if (refl && gt) DW_BWD(float, true, true); else if (refl) DW_BWD(float, true, false);
      else if (gt) DW_BWD(float, false, true); else DW_BWD(float, false, false);
    }
#undef DW_BWD
    const long W = (long)D * (k + 1);
    // (dw == NULL: the partial rows [gy][D][k + 1] stay in the workspace for a deferred smx_reduce_jobs)
    if (dw) hipLaunchKernelGGL(dw_partials_reduce_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, partial, (int)gy, D, k, dw, dbias);
  } else if (!dw) {
    return fail(SMX_EUNSUPPORTED, "smx_dwconv1d_glu_bwd: dw == NULL (deferred reduction) needs the k = 31 vector path");
  } else {
    // generic shapes (k != 31, D % 8 != 0, ...): with a workspace one partial row per workgroup + the fixed-order reduction
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, p, tiles_t, partial);
    else hipLaunchKernelGGL((dwconv_bwd_kernel<float>), grid, dim3(256), 0, s, p, tiles_t, partial);
    const long W = (long)D * (k + 1);
    if (partial) hipLaunchKernelGGL(dw_partials_reduce_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, partial, (int)gy, D, k, dw, dbias);
  }
  return check_launch("smx_dwconv1d_glu_bwd");
}
