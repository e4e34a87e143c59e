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
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwParams p, int tiles_t) {
  __shared__ float u[DW_ROWS][DW_CT];
  __shared__ float g[DW_ROWS][DW_CT];     // dYg with halo
  __shared__ float wl[DW_KMAX][DW_CT];
  __shared__ float dwl[DW_KMAX][DW_CT];
  __shared__ float dbl[DW_CT];
  const int cl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ch = blockIdx.x * DW_CT + cl;
  const int pad = (p.k - 1) / 2;
  const int rows = DW_TT + 2 * pad;
  for (int j = wv; j < p.k; j += 4) { wl[j][cl] = ch < p.D ? p.w[(long)ch * p.k + j] : 0.f; dwl[j][cl] = 0.f; }
  if (wv == 0) dbl[cl] = 0.f;
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
      for (int j = 0; j < p.k; ++j) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          int t = t0 + f0 + o, tau = t + j - pad;
          bool ok = p.chunk > 0 ? (tau < (t / p.chunk + 1) * p.chunk) : true;
          s += ok ? g[pad + f0 + o][cl] * u[f0 + o + j][cl] : 0.f;
        }
        atomicAdd(&dwl[j][cl], s);
      }
#pragma unroll
      for (int o = 0; o < 16; ++o) dbp += g[pad + f0 + o][cl];
      atomicAdd(&dbl[cl], dbp);
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
  __syncthreads();
  if (ch < p.D) {
    for (int j = wv; j < p.k; j += 4) atomicAdd(p.dw + (long)ch * p.k + j, dwl[j][cl]);
    if (wv == 0 && p.dbias) atomicAdd(p.dbias + ch, dbl[cl]);
  }
}

}  // namespace smx

using namespace smx;

extern "C" int smx_dwconv1d_glu_fwd(int dtype, const void* P, int64_t ldp, const float* w, const float* bias,
                                    const void* gate, int64_t ldg, void* Y, int64_t ldy, int B, int T, int D, int k,
                                    int glu, int pad_mode, int chunk, void* stream) {
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
  if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((dwconv_fwd_kernel<float>), grid, dim3(256), 0, s, p);
  return check_launch("smx_dwconv1d_glu_fwd");
}

extern "C" int smx_dwconv1d_glu_bwd(int dtype, const void* dY, int64_t lddy, const void* P, int64_t ldp, const float* w,
                                    const float* bias, const void* gate, int64_t ldg, void* dP, int64_t lddp,
                                    void* dgate, int64_t lddg, float* dw, float* dbias, int B, int T, int D, int k,
                                    int glu, int pad_mode, int chunk, void* stream) {
  SMX_REQUIRE(dY && P && w && dP && dw, "smx_dwconv1d_glu_bwd: null pointer");
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
  if (dtype == SMX_BF16) hipLaunchKernelGGL((dwconv_bwd_kernel<bf16_t>), grid, dim3(256), 0, s, p, tiles_t);
  else hipLaunchKernelGGL((dwconv_bwd_kernel<float>), grid, dim3(256), 0, s, p, tiles_t);
  return check_launch("smx_dwconv1d_glu_bwd");
}
