// rowwise.hip — HBM-bound row-wise kernels of the SummaryMixing path (gfx950, wave64):
//   masked mean over time (split-T, fixed-order combine) + broadcast backward, DynChunk window mean,
//   LayerNorm fwd/bwd, activation/mask backward with fused bias / per-utterance column sums,
//   axpby, casts, fused AdamW, sum of squares, clip factor.
// Every kernel moves 8-16 bytes per lane per access along the feature dim (coalesced 512 B - 1 KiB per
// wave instruction) and accumulates in fp32.
#include <stdlib.h>

#include "smx_common.h"

namespace smx {

template <typename T> struct VT;                       // elements per 16-byte lane access
template <> struct VT<float>  { static constexpr int N = 4; };
template <> struct VT<bf16_t> { static constexpr int N = 8; };

template <typename T, bool VEC>
__device__ __forceinline__ void loadv(const T* p, int nvalid, float (&f)[VT<T>::N]) {
  constexpr int N = VT<T>::N;
  if constexpr (VEC) {
    // whole aligned vectors only (host-checked): callers never issue this for an out-of-range column block
    if constexpr (sizeof(T) == 2) {
      uint4 r = *reinterpret_cast<const uint4*>(p);
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_bits_to_f32(w[i] & 0xffffu); f[2 * i + 1] = bf16_bits_to_f32(w[i] >> 16); }
    } else {
      float4 r = *reinterpret_cast<const float4*>(p);
      f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = (i < nvalid) ? to_f32(p[i]) : 0.f;
  }
}
template <typename T, bool VEC>
__device__ __forceinline__ void storev(T* p, int nvalid, const float (&f)[VT<T>::N]) {
  constexpr int N = VT<T>::N;
  if (VEC && nvalid >= N) {
    if constexpr (sizeof(T) == 2) {
      uint4 r;
      r.x = pack_bf16x2(f[0], f[1]);
      r.y = pack_bf16x2(f[2], f[3]);
      r.z = pack_bf16x2(f[4], f[5]);
      r.w = pack_bf16x2(f[6], f[7]);
      *reinterpret_cast<uint4*>(p) = r;
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < nvalid) p[i] = from_f32<T>(f[i]);
  }
}

static inline bool vec_ok(const void* p, int64_t ld, int D, int n, size_t es) {
  return p == nullptr || (aligned16(p) && ld % n == 0 && D % n == 0 && (ld * es) % 16 == 0);
}

// =================================================================================================
// masked mean over time.  stage 1: grid (DC, TS, B); 4 waves of a block interleave over the rows of the
// block's time range; each lane owns VT<T>::N consecutive features (16-byte loads, 4 rows in flight).
// =================================================================================================
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void masked_sum_stage1(const T* S, long lds, const uint8_t* mask, float* partial,
                                                         float* pcount, int T_, int D, int TR, int TS) {
  constexpr int N = VT<T>::N;
  __shared__ float red[3][64 * N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.z, ts = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * N;
  const int nvalid = D - col;
  const int t0 = ts * TR, t1 = min(T_, t0 + TR);
  float acc[N];
  float cnt = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.f;
  const T* base = S + ((long)b * T_) * lds + (nvalid > 0 ? col : 0);   // idle lanes re-read column 0 (discarded)
  const uint8_t* mrow = mask ? mask + (long)b * T_ : nullptr;
  // every pass keeps 8 independent 16-byte loads in flight per lane (128 B/lane, 8 KiB/wave); rows past the end of
  // the block's range re-read its last row with weight 0, so there is no scalar tail
  const int tlast = t1 - 1;
  for (int t = t0 + w; t < t1; t += 32) {
    float f[8][N], m[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) loadv<T, VEC>(base + (long)min(t + 4 * u, tlast) * lds, nvalid, f[u]);
    {
      const int tt = t + 4 * (lane & 7);                 // lane u (0..7) fetches the mask byte of row slot u
      const float mv = tt < t1 ? (mrow ? (mrow[tt] ? 1.f : 0.f) : 1.f) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) { m[u] = __shfl(mv, u, 64); cnt += m[u]; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc[i] += ((f[0][i] * m[0] + f[1][i] * m[1]) + (f[2][i] * m[2] + f[3][i] * m[3])) +
                ((f[4][i] * m[4] + f[5][i] * m[5]) + (f[6][i] * m[6] + f[7][i] * m[7]));
  }
  __shared__ float cred[4];
  if (lane == 0) cred[w] = cnt;
  if (w > 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[w - 1][lane * N + i] = acc[i];
  }
  __syncthreads();
  if (w == 0 && nvalid > 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = ((acc[i] + red[0][lane * N + i]) + red[1][lane * N + i]) + red[2][lane * N + i];
    float* o = partial + ((long)b * TS + ts) * D + col;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < nvalid) o[i] = acc[i];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) pcount[(long)b * TS + ts] = (cred[0] + cred[1]) + (cred[2] + cred[3]);
}

// stage 2: fixed-order sum of the TS partials, divide by the number of valid frames.  grid (ceil(D/256), B)
__global__ __launch_bounds__(256) void masked_sum_stage2(const float* partial, const float* pcount, float* out,
                                                         float* inv_count, int T_, int D, int TS, int scale) {
  const int b = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
  float c = 0.f;
  for (int ts = 0; ts < TS; ++ts) c += pcount[(long)b * TS + ts];   // every thread sums the same TS counts (L1 hits)
  const float inv = 1.f / c;   // zero valid frames -> inf, 0*inf = NaN, as the reference (summary_mixing.py:264-266)
  if (inv_count && blockIdx.x == 0 && threadIdx.x == 0) inv_count[b] = inv;
  if (col >= D) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int ts = 0;
  for (; ts + 3 < TS; ts += 4) {
    s0 += partial[((long)b * TS + ts) * D + col]; s1 += partial[((long)b * TS + ts + 1) * D + col];
    s2 += partial[((long)b * TS + ts + 2) * D + col]; s3 += partial[((long)b * TS + ts + 3) * D + col];
  }
  for (; ts < TS; ++ts) s0 += partial[((long)b * TS + ts) * D + col];
  const float s = (s0 + s1) + (s2 + s3);
  out[(long)b * D + col] = scale ? s * inv : s;
}

static void mm_plan(int B, int T, int D, int nvec, int& DC, int& TS, int& TR) {
  DC = (D + 64 * nvec - 1) / (64 * nvec);
  const long target = 512;                                 // workgroup target (swept 256 / 512 / 1024 / 2048: +-0.1 % of a step)
  long want = (target + (long)B * DC - 1) / ((long)B * DC);   // ~1024 workgroups (4 per CU) when the rows allow it
  long maxts = (T + 127) / 128;                             // >= 128 rows per block (32 per wave)
  TS = (int)(want < 1 ? 1 : (want > maxts ? maxts : want));
  if (TS < 1) TS = 1;
  TR = (T + TS - 1) / TS;
  TS = (T + TR - 1) / TR;
  if (TS < 1) TS = 1;
}

// dS[b,t,:] = g[b,:] * inv_count[b]   (broadcast over t).  grid (DC, ceil(T/RPB), B)
// Optional fusions: the dropout of the merge input (forward `repeat`), or - in the backward - the activation / mask
// backward of the projection that produced the summary columns: dS = g * inv * act'(Z[b,t,:]) * mask[b,t], so that the
// broadcast gradient is never written and re-read by a separate act_mask_bwd pass.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* g, const float* inv_count, T* dS, long ldds,
                                                         int T_, int D, int RPB, uint32_t dthresh, float dscale,
                                                         uint64_t dseed_, const uint64_t* ep, const T* __restrict__ Z, long ldz,
                                                         const uint8_t* __restrict__ mask, int act, int LPR) {
  const uint64_t dseed = dthresh ? epoch_seed(dseed_, ep) : 0;
  constexpr int N = VT<T>::N;
  // LPR lanes cover the columns of a row (a power of two <= 64): for narrow rows (D = 256 in bf16: 32 lanes) a wave
  // writes 64 / LPR rows per pass instead of leaving half its lanes idle
  const int lane = threadIdx.x & 63, b = blockIdx.z;
  const int RPW = 64 / LPR, w = (threadIdx.x >> 6) * RPW + lane / LPR, WS = 4 * RPW;
  const int col = (blockIdx.x * LPR + (lane & (LPR - 1))) * N;
  const int nvalid = D - col;
  if (nvalid <= 0) return;
  float v[N];
  const float sc = inv_count ? inv_count[b] : 1.f;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (i < nvalid) ? g[(long)b * D + col + i] * sc : 0.f;
  const int t0 = blockIdx.y * RPB, t1 = min(T_, t0 + RPB);
  if (Z || mask) {                                       // backward through act(.) * mask of the producing projection
    dispatch_act(Z ? act : SMX_ACT_NONE, [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      for (int t = t0 + w; t < t1; t += WS) {
        const long n = (long)b * T_ + t;
        const float mk = mask ? (mask[n] ? 1.f : 0.f) : 1.f;
        float o[N];
        if (ACT != SMX_ACT_NONE) {
          float z[N];
          loadv<T, VEC>(Z + n * ldz + col, nvalid, z);
#pragma unroll
          for (int i = 0; i < N; ++i) o[i] = v[i] * mk * act_grad_c<ACT>(z[i]);
        } else {
#pragma unroll
          for (int i = 0; i < N; ++i) o[i] = v[i] * mk;
        }
        storev<T, VEC>(dS + n * ldds + col, nvalid, o);
      }
    });
  } else if (dthresh == 0) {
    for (int t = t0 + w; t < t1; t += WS) storev<T, VEC>(dS + ((long)b * T_ + t) * ldds + col, nvalid, v);
  } else {                                               // fused inverted dropout, mask = f(seed, row * D + col)
    for (int t = t0 + w; t < t1; t += WS) {
      const uint64_t base = ((uint64_t)b * T_ + t) * (uint64_t)D + col;
      float o[N];
#pragma unroll
      for (int i = 0; i < N; ++i) o[i] = v[i];
      dropout_apply_any<N>(o, dseed, base, dthresh, dscale);
      storev<T, VEC>(dS + ((long)b * T_ + t) * ldds + col, nvalid, o);
    }
  }
}

// =================================================================================================
// DynChunk window mean (O(T)).  chunk sums -> window combine.
// =================================================================================================
// csum[b,c,:] = scale_c * sum_{t in chunk c} X[b,t,:]   (scale_c = 1 or 1/wlen(c)).  grid (DC, NC, B)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void chunk_sum_kernel(const T* X, long ldx, float* csum, int T_, int D, int chunk,
                                                        int NC, int left, int scale_by_wlen, int c_off) {
  constexpr int N = VT<T>::N;
  __shared__ float red[3][64 * N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.z, c = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * N;
  const int nvalid = D - col;
  const int t0 = c * chunk, t1 = min(T_, t0 + chunk);
  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.f;
  for (int t = t0 + w; t < t1; t += 4) {
    float f[N];
    loadv<T, VEC>(X + ((long)b * T_ + t) * ldx + (nvalid > 0 ? col : 0), nvalid, f);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] += f[i];
  }
  if (w > 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[w - 1][lane * N + i] = acc[i];
  }
  __syncthreads();
  if (w == 0 && nvalid > 0) {
    float sc = 1.f;
    if (scale_by_wlen) {                                  // (c_off: chunks of earlier sequence-parallel shards in front of this one)
      const int lo = left < 0 ? 0 : max(0, (c + c_off - left) * chunk);
      sc = 1.f / (float)(t1 + c_off * chunk - lo);
    }
    float* o = csum + ((long)b * NC + c) * D + col;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < nvalid) o[i] = sc * (((acc[i] + red[0][lane * N + i]) + red[1][lane * N + i]) + red[2][lane * N + i]);
  }
}
// Unlimited left context: the window of chunk c is the whole prefix 0..c (reverse: the suffix c..NC-1), which the window
// kernel used to add up chunk by chunk - a dependent chain of up to NC loads per workgroup (237 us at NC = 32 against ~10 us of
// bytes).  One pass turns csum into its running sums in place instead (thread = one column of one utterance, the NC loads
// are independent and unrolled), and the window kernel reads ONE row.  Fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void chunk_prefix_kernel(float* csum, int D, int NC, int reverse) {
  const int col = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (col >= D) return;
  float* p = csum + (long)b * NC * D + col;
  float acc = 0.f;
  if (!reverse) {
#pragma unroll 8
    for (int c = 0; c < NC; ++c) { acc += p[(long)c * D]; p[(long)c * D] = acc; }
  } else {
#pragma unroll 8
    for (int c = NC - 1; c >= 0; --c) { acc += p[(long)c * D]; p[(long)c * D] = acc; }
  }
}
// fwd: out rows of chunk c = (sum_{c'=lo..c} csum[c']) / wlen(c);  bwd (reverse=1): rows of chunk c =
// sum_{c'=c..hi} csum[c'] (csum already scaled by 1/wlen).  grid (DC, NC, B)
// Sequence-parallel shards (c_off chunks in front of this one; carry): the window sums that reach into other shards arrive as
// `carry` rows - chunk c adds carry[b][c - carry_c0] for carry_c0 <= c < carry_c0 + carry_n (carry_n == 0 with a non-null carry: ONE
// row per utterance for every chunk: the totals of all earlier / later shards with unlimited left context).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void chunk_window_kernel(const float* csum, T* out, long ldo, int T_, int D,
                                                           int chunk, int NC, int left, int reverse, int c_off,
                                                           const float* __restrict__ carry, int carry_c0, int carry_n) {
  constexpr int N = VT<T>::N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.z, c = blockIdx.y;
  const int col = (blockIdx.x * 64 + lane) * N;
  const int nvalid = D - col;
  if (nvalid <= 0) return;
  int clo, chi;
  if (left < 0) { clo = c; chi = c; }                    // (csum holds running sums already: chunk_prefix_kernel)
  else if (!reverse) { clo = max(0, c - left); chi = c; }
  else { clo = c; chi = min(NC - 1, c + left); }
  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.f;
  for (int cc = clo; cc <= chi; ++cc) {
    const float* p = csum + ((long)b * NC + cc) * D + col;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < nvalid) acc[i] += p[i];
  }
  if (carry) {
    const float* cp = nullptr;
    if (carry_n == 0) cp = carry + (long)b * D + col;
    else if (c >= carry_c0 && c < carry_c0 + carry_n) cp = carry + ((long)b * carry_n + (c - carry_c0)) * D + col;
    if (cp) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i < nvalid) acc[i] += cp[i];
    }
  }
  const int t0 = c * chunk, t1 = min(T_, t0 + chunk);
  if (!reverse) {
    const int lo = left < 0 ? 0 : max(0, (c + c_off - left) * chunk);
    float sc = 1.f / (float)(t1 + c_off * chunk - lo);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] *= sc;
  }
  for (int t = t0 + w; t < t1; t += 4) storev<T, VEC>(out + ((long)b * T_ + t) * ldo + col, nvalid, acc);
}

// =================================================================================================
// SummaryMixing-expdecay in O(T).  The reference builds the (T,T) Laplace matrix
// M_ij = gamma^|i-j| (summary_mixing.py:316-365) and evaluates (M s) / rowsum(M) (:233-235): O(T^2).  M s is a two-sided
// exponential filter:  f_t = s_t + gamma f_{t-1},  g_t = s_t + gamma g_{t+1},  (M s)_t = f_t + g_t - s_t,  and
// rowsum(M)_t = (2 - gamma^(t+1) - gamma^(T-t)) / (1 - gamma) - 1 in closed form (the denominator ignores padding like
// the reference's).  Chunked scan over T: (1) per 16-row chunk the local end values of both recurrences, (2) carries
// across the chunks (a short sequential loop per column), (3) per chunk both local scans out of registers + carries.
// mode 0 (forward):  out = (M s) / rowsum(M);   mode 1 (backward, M symmetric): out = M (s / rowsum(M)).
// =================================================================================================
constexpr int ED_CH = 16;                                // rows per chunk

__device__ __forceinline__ float ed_inv_den(int t, int T_, float lng, float inv1mg) {
  const float den = (2.f - expf((float)(t + 1) * lng) - expf((float)(T_ - t) * lng)) * inv1mg - 1.f;
  return 1.f / den;
}

// grid (DC, ceil(NC/4), B); a wave owns one chunk; lane owns 4 columns.  ws[b][c][0][D] = F_c, ws[b][c][1][D] = G_c
template <typename T>
__global__ __launch_bounds__(256) void expdecay_chunk_kernel(const T* __restrict__ X, long ldx, float* __restrict__ ws,
                                                             int T_, int D, int NC, float gamma, int mode, int t_off, int T_glob) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.z;
  const int c = blockIdx.y * 4 + w, col = (blockIdx.x * 64 + lane) * 4;
  if (c >= NC || col >= D) return;
  const float lng = logf(gamma), inv1mg = 1.f / (1.f - gamma);
  const int t0 = c * ED_CH;
  float F[4] = {0.f, 0.f, 0.f, 0.f}, G[4] = {0.f, 0.f, 0.f, 0.f};
  float v[ED_CH][4];
#pragma unroll
  for (int i = 0; i < ED_CH; ++i) {
    const int t = t0 + i;
    if (t < T_) load4<T>(X + ((long)b * T_ + t) * ldx + col, v[i]);
    else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
  }
  float pw = 1.f;
#pragma unroll
  for (int i = 0; i < ED_CH; ++i) {
    const int t = t0 + i;
    const float sc = (mode == 1 && t < T_) ? ed_inv_den(t + t_off, T_glob, lng, inv1mg) : 1.f;   // (t_off / T_glob: a sequence-parallel shard)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x = v[i][q] * sc;
      F[q] = x + gamma * F[q];                           // = sum_t gamma^(last - t) x_t  (rows past T count as zeros)
      G[q] += pw * x;                                    // = sum_t gamma^(t - first) x_t
    }
    pw *= gamma;
  }
  float* o = ws + (((long)b * NC + c) * 2) * D + col;
  *reinterpret_cast<float4*>(o) = make_float4(F[0], F[1], F[2], F[3]);
  *reinterpret_cast<float4*>(o + D) = make_float4(G[0], G[1], G[2], G[3]);
}

// in place: ws[b][c][0] <- f entering chunk c from the left, ws[b][c][1] <- g entering chunk c from the right.
// One thread per (b, column); the chunk values are independent of the recurrence, so 8 are fetched ahead.
// f_in / g_in (B, D; sequence-parallel shards): the states entering the shard from its left / right neighbours, else zero.
// ends (2, B, D; optional, then NOTHING is written to ws): the states LEAVING the shard - f at its last frame, g at its first -
// with zero entering states: what the shards exchange before the second call.
__global__ __launch_bounds__(256) void expdecay_carry_kernel(float* __restrict__ ws, int D, int NC, int B, float gamma,
                                                             const float* __restrict__ f_in, const float* __restrict__ g_in,
                                                             float* __restrict__ ends, int T_) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)B * D) return;
  const int b = (int)(idx / D), col = (int)(idx % D);
  float* base = ws + (long)b * NC * 2 * D + col;
  const float gch = expf((float)ED_CH * logf(gamma));    // every chunk is ED_CH rows long (the tail is zero padded)
  // both directions advance in the same loop (two independent chains), 16 chunk values of each fetched ahead
  float cf = f_in ? f_in[idx] : 0.f, cg = g_in ? g_in[idx] : 0.f;
  for (int c0 = 0; c0 < NC; c0 += 16) {
    float f[16], g[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = c0 + u;
      f[u] = c < NC ? base[(long)c * 2 * D] : 0.f;
      g[u] = c < NC ? base[(long)(NC - 1 - c) * 2 * D + D] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = c0 + u;
      if (c < NC) {
        if (!ends) base[(long)c * 2 * D] = cf;
        cf = f[u] + gch * cf;
        if (!ends) base[(long)(NC - 1 - c) * 2 * D + D] = cg;
        cg = g[u] + gch * cg;
      }
    }
  }
  if (ends) {
    // the last chunk is zero padded to ED_CH rows: its end value is gamma^pad times f at the shard's last frame
    const int pad = NC * ED_CH - T_;
    ends[idx] = cf * expf(-(float)pad * logf(gamma));
    ends[(long)B * D + idx] = cg;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void expdecay_apply_kernel(const T* __restrict__ X, long ldx, const float* __restrict__ ws,
                                                             T* __restrict__ Y, long ldy, int T_, int D, int NC, float gamma,
                                                             int mode, int t_off, int T_glob) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.z;
  const int c = blockIdx.y * 4 + w, col = (blockIdx.x * 64 + lane) * 4;
  if (c >= NC || col >= D) return;
  const float lng = logf(gamma), inv1mg = 1.f / (1.f - gamma);
  const int t0 = c * ED_CH;
  float v[ED_CH][4], a[ED_CH][4];
#pragma unroll
  for (int i = 0; i < ED_CH; ++i) {
    const int t = t0 + i;
    if (t < T_) load4<T>(X + ((long)b * T_ + t) * ldx + col, v[i]);
    else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
  }
  const float* cw = ws + (((long)b * NC + c) * 2) * D + col;
  const float4 cf4 = *reinterpret_cast<const float4*>(cw), cg4 = *reinterpret_cast<const float4*>(cw + D);
  float f[4] = {cf4.x, cf4.y, cf4.z, cf4.w}, g[4] = {cg4.x, cg4.y, cg4.z, cg4.w};
#pragma unroll
  for (int i = 0; i < ED_CH; ++i) {                      // ascending: a_t = gamma f_{t-1}  (= f_t - x_t)
    const int t = t0 + i;
    const float sc = (mode == 1 && t < T_) ? ed_inv_den(t + t_off, T_glob, lng, inv1mg) : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[i][q] *= sc;
      a[i][q] = gamma * f[q];
      f[q] = v[i][q] + a[i][q];
    }
  }
#pragma unroll
  for (int i = ED_CH - 1; i >= 0; --i) {                 // descending: g_t = x_t + gamma g_{t+1};  out = a_t + g_t
    const int t = t0 + i;
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g[q] = v[i][q] + gamma * g[q];
      o[q] = a[i][q] + g[q];
    }
    if (t < T_) {
      if (mode == 0) {
        const float sc = ed_inv_den(t + t_off, T_glob, lng, inv1mg);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] *= sc;
      }
      store4<T>(Y + ((long)b * T_ + t) * ldy + col, o);
    }
  }
}

// =================================================================================================
// LayerNorm.  One wave per row, 4 rows per block; lane owns 4-element vectors at columns lane*4 + 256*i.
// =================================================================================================
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* X, long ldx, const float* gamma, const float* beta,
                                                            T* Y, long ldy, float* stats, int N_, int D, float eps,
                                                            int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N_) return;
  const T* x = X + (long)row * ldx;
  T* y = Y + (long)row * ldy;
  float s = 0.f;
  if (VEC) {
    for (int c = lane * 4; c < D; c += 256) { float f[4]; load4<T>(x + c, f); s += (f[0] + f[1]) + (f[2] + f[3]); }
  } else {
    for (int c = lane; c < D; c += 64) s += to_f32(x[c]);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  if (VEC) {
    for (int c = lane * 4; c < D; c += 256) {
      float f[4]; load4<T>(x + c, f);
#pragma unroll
      for (int i = 0; i < 4; ++i) { float d = f[i] - mean; q += d * d; }
    }
  } else {
    for (int c = lane; c < D; c += 64) { float d = to_f32(x[c]) - mean; q += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (stats && lane == 0) { stats[2 * (long)row] = mean; stats[2 * (long)row + 1] = rstd; }
  if (VEC) {
    for (int c = lane * 4; c < D; c += 256) {
      float f[4]; load4<T>(x + c, f);
      float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
      float o[4] = {(f[0] - mean) * rstd * g4.x + b4.x, (f[1] - mean) * rstd * g4.y + b4.y,
                    (f[2] - mean) * rstd * g4.z + b4.z, (f[3] - mean) * rstd * g4.w + b4.w};
      if (act == SMX_ACT_SWISH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_fwd_c<SMX_ACT_SWISH>(o[i]);
      } else if (act == SMX_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_fwd_c<SMX_ACT_GELU>(o[i]);
      } else if (act != SMX_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_fwd(act, o[i]);
      }
      store4<T>(y + c, o);
    }
  } else {
    for (int c = lane; c < D; c += 64) y[c] = from_f32<T>(act_fwd(act, (to_f32(x[c]) - mean) * rstd * gamma[c] + beta[c]));
  }
}

// Single-read forward for D <= 256 * CH (D % 4 == 0, aligned rows): the row lives in registers (one 4-element vector
// per lane and chunk), U rows are in flight per wave (all their loads issued before the first reduction), workgroups
// stride over the rows.  The generic kernel above re-reads the row three times behind three dependent latencies.
// TX: element type of the input (float for the fp32 residual stream: LayerNorm(fp32 x) -> dtype T, smx_layernorm_fwd_x32)
template <typename T, int CH, int U, typename TX = T>
__global__ __launch_bounds__(256) void layernorm_fwd_fast(const TX* __restrict__ X, long ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ Y, long ldy,
                                                          float* __restrict__ stats, int N_, int D, float eps, int act) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float gam[CH][4], bet[CH][4];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
      gam[i][0] = g4.x; gam[i][1] = g4.y; gam[i][2] = g4.z; gam[i][3] = g4.w;
      bet[i][0] = b4.x; bet[i][1] = b4.y; bet[i][2] = b4.z; bet[i][3] = b4.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) gam[i][j] = bet[i][j] = 0.f;
    }
  }
  const float invD = 1.f / (float)D;
  dispatch_act(act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    for (int row0 = (blockIdx.x * 4 + w) * U; row0 < N_; row0 += gridDim.x * 4 * U) {
      float f[U][CH][4], s[U], q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + u, N_ - 1);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * 4;
          if (c < D) load4<TX>(X + (long)row * ldx + c, f[u][i]);
          else f[u][i][0] = f[u][i][1] = f[u][i][2] = f[u][i][3] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) s[u] += (f[u][i][0] + f[u][i][1]) + (f[u][i][2] + f[u][i][3]);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] += __shfl_xor(s[u], off, 64);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] *= invD;                                      // mean
        q[u] = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if ((lane + 64 * i) * 4 < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = f[u][i][j] - s[u]; q[u] += d * d; }
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] += __shfl_xor(q[u], off, 64);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + u;
        if (row >= N_) break;
        const float rstd = rsqrtf(q[u] * invD + eps);
        if (stats && lane == 0) *reinterpret_cast<float2*>(stats + 2 * (long)row) = make_float2(s[u], rstd);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * 4;
          if (c < D) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_fwd_c<ACT>((f[u][i][j] - s[u]) * rstd * gam[i][j] + bet[i][j]);
            store4<T>(Y + (long)row * ldy + c, o);
          }
        }
      }
    }
  });
}

// Two LayerNorms in one pass over the float32 residual stream (round 4): y1 = LN1(x) (float32: the layer-final norm2 of
// a Conformer layer, Conformer.py:536 = the next layer's stream input) and y2 = LN2(y1) (dtype T2: the LayerNorm in front of
// the next layer's first feed-forward module, Conformer.py:458-459,507).  y1 never comes back from memory for the second
// statistics.  Same lane / chunk layout and the same reduction trees as layernorm_fwd_fast, so both outputs equal those of
// two separate launches to an ulp.
template <typename T2, int CH, int U>
__global__ __launch_bounds__(256) void layernorm_fwd_pair_fast(const float* __restrict__ X, long ldx, const float* __restrict__ gamma1,
                                                               const float* __restrict__ beta1, float eps1, float* __restrict__ Y1,
                                                               long ldy1, float* __restrict__ stats1,
                                                               const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                               float eps2, T2* __restrict__ Y2, long ldy2, float* __restrict__ stats2,
                                                               int N_, int D) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g1[CH][4], b1[CH][4], g2[CH][4], b2[CH][4];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (lane + 64 * i) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) g1[i][j] = b1[i][j] = g2[i][j] = b2[i][j] = 0.f;
    if (c < D) {
      load4<float>(gamma1 + c, g1[i]); load4<float>(beta1 + c, b1[i]);
      load4<float>(gamma2 + c, g2[i]); load4<float>(beta2 + c, b2[i]);
    }
  }
  const float invD = 1.f / (float)D;
  auto row_sum = [&](float (&v)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] += __shfl_xor(v[u], off, 64);
  };
  for (int row0 = (blockIdx.x * 4 + w) * U; row0 < N_; row0 += gridDim.x * 4 * U) {
    float f[U][CH][4], s[U], q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = min(row0 + u, N_ - 1);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) load4<float>(X + (long)row * ldx + c, f[u][i]);
        else f[u][i][0] = f[u][i][1] = f[u][i][2] = f[u][i][3] = 0.f;
      }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {               // pass 0: LN1 (f <- y1, stored), pass 1: LN2 of the registers
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) s[u] += (f[u][i][0] + f[u][i][1]) + (f[u][i][2] + f[u][i][3]);
      }
      row_sum(s);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] *= invD;
        q[u] = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if ((lane + 64 * i) * 4 < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = f[u][i][j] - s[u]; q[u] += d * d; }
          }
        }
      }
      row_sum(q);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + u;
        const bool live = row < N_;
        const float rstd = rsqrtf(q[u] * invD + (pass == 0 ? eps1 : eps2));
        float* st = pass == 0 ? stats1 : stats2;
        if (live && st && lane == 0) *reinterpret_cast<float2*>(st + 2 * (long)row) = make_float2(s[u], rstd);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * 4;
          if (c < D) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              o[j] = (f[u][i][j] - s[u]) * rstd * (pass == 0 ? g1[i][j] : g2[i][j]) + (pass == 0 ? b1[i][j] : b2[i][j]);
            if (pass == 0) {
              if (live) {
                // non-temporal: the stream tensor is next read by a residual epilogue several kernels later, Y2 by the very next
                // GEMM - with an ordinary store the 131 MB of Y1 pushed Y2 out of the 256 MB Infinity Cache at 64000 x 512
                // (that GEMM 220 -> 257 us, the C2a step +0.35 ms; with the hint -0.35 ms against two launches)
                typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
                u32x4n uu = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                __builtin_nontemporal_store(uu, reinterpret_cast<u32x4n*>(Y1 + (long)row * ldy1 + c));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) f[u][i][j] = o[j];
            } else if (live) {
              store4<T2>(Y2 + (long)row * ldy2 + c, o);
            }
          }
        }
      }
    }
  }
}

// bwd: dx = R + rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*act'(LN(x))*gamma.  Blocks stride over rows;
// gamma/beta of the lane's columns live in registers for the whole kernel, U rows are in flight per wave (all
// their loads issued before any reduction), dgamma/dbeta partial sums stay in registers until one atomic flush.
// Optional second output of the LayerNorm backward: dX2 = alpha * Dropout(dX; seed) * row_mask - the first thing the
// NEXT backward block does to this gradient (FFN: 1/2 * D(dy), conv module: D(dy) * mask).  Written from the registers
// that hold dX anyway, it replaces a separate elementwise pass (one more read and one more launch per module).
struct LnSecond {
  void* dX2; long ld; float alpha; const uint8_t* mask; uint32_t thresh; float scale; uint64_t seed; const uint64_t* epoch;
  // round 6 (split-K dgrads of a small batch, smx_gemm_panel_slabs): the incoming gradient dY = the sum of `nslab` float32 slabs
  // ((N, D) each, `slab_stride` elements apart, added in slab order) instead of a dtype-T tensor; null: dY as given
  const float* slabs; int nslab; long slab_stride;
};

template <typename T, int VW, int CH, int U, typename TX = T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dY, long lddy, const TX* __restrict__ X,
                                                            long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act,
                                                            const float* __restrict__ stats, const T* __restrict__ R,
                                                            long ldr, T* __restrict__ dX, long lddx,
                                                            float* __restrict__ partial, int N_, int D, LnSecond sec) {
  const uint64_t sseed = sec.dX2 ? epoch_seed(sec.seed, sec.epoch) : 0;
  __shared__ float red[3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float gam[CH][VW], bet[CH][VW], dg[CH][VW], db[CH][VW];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int c = (lane + 64 * i) * VW + j;
      gam[i][j] = c < D ? gamma[c] : 0.f;
      bet[i][j] = (c < D && act != SMX_ACT_NONE) ? beta[c] : 0.f;
      dg[i][j] = db[i][j] = 0.f;
    }
  dispatch_act(act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    for (int row0 = (blockIdx.x * 4 + w) * U; row0 < N_; row0 += gridDim.x * 4 * U) {
      float fdy[U][CH][VW], fx[U][CH][VW], fr[U][CH][VW == 4 ? 4 : 1], mean[U], rstd[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + u, N_ - 1);           // tail rows re-read the last row (results discarded)
        mean[u] = stats[2 * (long)row];
        rstd[u] = stats[2 * (long)row + 1];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * VW;
          const int cc = c < D ? c : 0;                  // idle lanes re-read column 0
          if constexpr (VW == 4) {
            if (sec.slabs) {                               // (uniform)
              const float* sp = sec.slabs + (long)row * D + cc;
              fdy[u][i][0] = fdy[u][i][1] = fdy[u][i][2] = fdy[u][i][3] = 0.f;
              for (int s0 = 0; s0 < sec.nslab; s0 += 4) {  // four slabs in flight, summed in slab order
                float4 a4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a4[k] = *reinterpret_cast<const float4*>(sp + (long)min(s0 + k, sec.nslab - 1) * sec.slab_stride);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (s0 + k < sec.nslab) { fdy[u][i][0] += a4[k].x; fdy[u][i][1] += a4[k].y; fdy[u][i][2] += a4[k].z; fdy[u][i][3] += a4[k].w; }
              }
            } else {
              load4<T>(dY + (long)row * lddy + cc, fdy[u][i]);
            }
            load4<TX>(X + (long)row * ldx + cc, fx[u][i]);
          }
          else { fdy[u][i][0] = to_f32(dY[(long)row * lddy + cc]); fx[u][i][0] = to_f32(X[(long)row * ldx + cc]); }
          // the residual gradient is requested with the operands: loaded after the reductions it was a second dependent
          // round trip per row group
          if constexpr (VW == 4) {
            if (R) load4<T>(R + (long)row * ldr + cc, fr[u][i]);
            else fr[u][i][0] = fr[u][i][1] = fr[u][i][2] = fr[u][i][3] = 0.f;
          }
        }
      }
      float s1[U], s2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool rok = row0 + u < N_;
        s1[u] = s2[u] = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const bool cok = (lane + 64 * i) * VW < D;
#pragma unroll
          for (int j = 0; j < VW; ++j) {
            const float xhat = (fx[u][i][j] - mean[u]) * rstd[u];
            float dyn = (cok && rok) ? fdy[u][i][j] : 0.f;
            if constexpr (ACT != SMX_ACT_NONE) dyn *= act_grad_c<ACT>(xhat * gam[i][j] + bet[i][j]);
            const float g = dyn * gam[i][j];
            fx[u][i][j] = xhat;          // reuse registers: fx <- xhat, fdy <- g
            fdy[u][i][j] = g;
            s1[u] += g;
            s2[u] += g * xhat;
            dg[i][j] += dyn * xhat;
            db[i][j] += dyn;
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) { s1[u] += __shfl_xor(s1[u], off, 64); s2[u] += __shfl_xor(s2[u], off, 64); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + u;
        if (row >= N_) continue;
        const float m1 = s1[u] / (float)D, m2 = s2[u] / (float)D;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * VW;
          if (c < D) {
            float o[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) o[j] = rstd[u] * (fdy[u][i][j] - m1 - fx[u][i][j] * m2);
            if constexpr (VW == 4) {
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] += fr[u][i][j];
            } else {
              if (R) o[0] += to_f32(R[(long)row * ldr + c]);
            }
            if constexpr (VW == 4) store4<T>(dX + (long)row * lddx + c, o);
            else dX[(long)row * lddx + c] = from_f32<T>(o[0]);
            if (sec.dX2) {                               // (uniform)
              const float mk = (sec.mask ? (sec.mask[row] ? 1.f : 0.f) : 1.f) * sec.alpha;
              if (sec.thresh) dropout_apply_any<VW>(o, sseed, (uint64_t)row * D + c, sec.thresh, sec.scale);
#pragma unroll
              for (int j = 0; j < VW; ++j) o[j] *= mk;
              T* d2 = reinterpret_cast<T*>(sec.dX2);
              if constexpr (VW == 4) store4<T>(d2 + (long)row * sec.ld + c, o);
              else d2[(long)row * sec.ld + c] = from_f32<T>(o[0]);
            }
          }
        }
      }
    }
  });
  // flush dgamma / dbeta: reduce the 4 waves through LDS and write ONE partial row per block (no atomics: with a
  // few thousand blocks adding into the same D addresses the L2 atomic unit serialised, 200 us per call).
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        float v = pass == 0 ? dg[i][j] : db[i][j];
        __syncthreads();
        if (w > 0) red[w - 1][lane] = v;
        __syncthreads();
        if (w == 0) {
          v = ((v + red[0][lane]) + red[1][lane]) + red[2][lane];
          const int c = (lane + 64 * i) * VW + j;
          if (c < D) partial[((long)blockIdx.x * 2 + pass) * D + c] = v;
        }
      }
  }
}

#ifndef SMX_LN_WG8_UF
#define SMX_LN_WG8_UF 2        // rows in flight per workgroup, forward
#endif
#ifndef SMX_LN_WG8_FBLOCKS
#define SMX_LN_WG8_FBLOCKS 2048
#endif
#ifndef SMX_LN_WG8_FROM
#define SMX_LN_WG8_FROM 1024   // rows wider than this (and <= 2048, bf16) take the workgroup-per-row kernels
#endif
// ---- mid-width rows (1024 < D <= 2048, bf16; the CSGU LayerNorm over 1536 channels of the Branchformer's cgMLP) ----------------
// One WORKGROUP per row, thread t owns the 8 consecutive columns 8 t (one 16-byte access per tensor and row), U rows in flight
// per iteration, workgroups stride over the rows; gamma / beta (and the dgamma / dbeta partial sums) of the thread's columns live
// in registers for the whole kernel.  The wave-per-row kernels above need 8 chunks of 4 columns per lane at this width: 128
// parameter registers per lane (backward: 256 VGPRs = ONE wave per SIMD with one 9 KB row in flight: 2.45 TB/s; forward: every
// one of the 8192 waves fetched its own 12 KB of gamma / beta for ~4 rows of 3 KB: 2.2 TB/s; tools/step_records.py c4).
__device__ __forceinline__ void ld8_bf16(const bf16_t* p, float (&f)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const float (&f)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__device__ __forceinline__ void ld8_f32(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// sums of U values per thread over the workgroup (4 waves): wave shuffles, then 4 partials per value through LDS
template <int U>
__device__ __forceinline__ void wg_sum(float (&v)[U], float (*red)[4], int lane, int w) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += __shfl_xor(v[u], off, 64);
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) red[u][w] = v[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = (red[u][0] + red[u][1]) + (red[u][2] + red[u][3]);
}

template <int U>
__global__ __launch_bounds__(256) void layernorm_fwd_wg8_kernel(const bf16_t* __restrict__ X, long ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ Y, long ldy,
                                                                float* __restrict__ stats, int N_, int D, float eps, int act) {
  __shared__ float red[2][U][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = t * 8;
  const bool in = c < D;
  float gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { gam[j] = in ? gamma[c + j] : 0.f; bet[j] = in ? beta[c + j] : 0.f; }
  const float invD = 1.f / (float)D;
  dispatch_act(act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    for (int row0 = blockIdx.x * U; row0 < N_; row0 += gridDim.x * U) {
      float f[U][8], s[U], q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + u, N_ - 1);            // tail rows re-read the last row (results discarded)
        if (in) ld8_bf16(X + (long)row * ldx + c, f[u]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[u][j] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) s[u] = ((f[u][0] + f[u][1]) + (f[u][2] + f[u][3])) + ((f[u][4] + f[u][5]) + (f[u][6] + f[u][7]));
      wg_sum<U>(s, red[0], lane, w);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] *= invD;                                      // mean
        q[u] = 0.f;
        if (in) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = f[u][j] - s[u]; q[u] += d * d; }
        }
      }
      wg_sum<U>(q, red[1], lane, w);                       // (red[0] is rewritten only after this barrier: no race)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = row0 + u;
        if (row >= N_) break;
        const float rstd = rsqrtf(q[u] * invD + eps);
        if (stats && t == 0) *reinterpret_cast<float2*>(stats + 2 * (long)row) = make_float2(s[u], rstd);
        if (in) {
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = act_fwd_c<ACT>((f[u][j] - s[u]) * rstd * gam[j] + bet[j]);
          st8_bf16(Y + (long)row * ldy + c, o);
        }
      }
    }
  });
}

// backward (see layernorm_bwd_kernel for the formulas); TX = float: the LayerNorm input is the fp32 residual stream.
// Software-pipelined over the rows of the workgroup: the operands of row i + 1 are requested before row i is reduced, and stay
// PACKED (the 16 bytes as loaded) until they are consumed - 4 registers per bf16 tensor and row instead of 8, unpacked once for
// the row sums and once more for the outputs.  With one row in flight and nothing prefetched the kernel ran at the latency
// bound of 4 workgroups x 9 KB per CU (4.7 TB/s plain, 3.6 TB/s with the extra Z stream of PRE).
// PRE (smx_layernorm_bwd_preact): the LayerNorm input is X = zact(Z); the kernel then emits the gradient w.r.t. Z,
// dX * zact'(Z), from the registers that hold dX - the consumer's activation-backward pass over this tensor is gone.
template <typename TX>
struct LnRaw {
  uint4 dy, r, z;
  uint4 x0, x1;                                            // (bf16 x: x0 only)
  float mean, rstd;
};
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <bool PRE, typename TX>
__global__ __launch_bounds__(256, 4) void layernorm_bwd_wg8_kernel(const bf16_t* __restrict__ dY, long lddy, const TX* __restrict__ X, long ldx,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                   const float* __restrict__ stats, const bf16_t* __restrict__ R, long ldr,
                                                                   bf16_t* __restrict__ dX, long lddx, float* __restrict__ partial, int N_, int D,
                                                                   const bf16_t* __restrict__ Zp, long ldz, int zact) {
  __shared__ float red[2][2][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = t * 8;
  const bool in = c < D;
  const int cc = in ? c : 0;                               // (idle threads re-read column 0; their results are masked)
  float gam[8], bet[8], dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gam[j] = in ? gamma[c + j] : 0.f;
    bet[j] = (in && act != SMX_ACT_NONE) ? beta[c + j] : 0.f;
    dg[j] = db[j] = 0.f;
  }
  const float invD = 1.f / (float)D;
  auto fetch = [&](int row, LnRaw<TX>& q) {
    q.mean = stats[2 * (long)row];
    q.rstd = stats[2 * (long)row + 1];
    q.dy = *reinterpret_cast<const uint4*>(dY + (long)row * lddy + cc);
    if constexpr (sizeof(TX) == 4) {
      const float* xp = reinterpret_cast<const float*>(X) + (long)row * ldx + cc;
      q.x0 = *reinterpret_cast<const uint4*>(xp);
      q.x1 = *reinterpret_cast<const uint4*>(xp + 4);
    } else {
      q.x0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(X) + (long)row * ldx + cc);
    }
    if constexpr (!PRE) {
      if (R) q.r = *reinterpret_cast<const uint4*>(R + (long)row * ldr + cc);   // (uniform; the PRE variant has no residual gradient)
    }
    if constexpr (PRE) q.z = *reinterpret_cast<const uint4*>(Zp + (long)row * ldz + cc);
  };
  auto xhat8 = [&](const LnRaw<TX>& q, float (&xh)[8]) {
    if constexpr (sizeof(TX) == 4) {
      const uint32_t wv[8] = {q.x0.x, q.x0.y, q.x0.z, q.x0.w, q.x1.x, q.x1.y, q.x1.z, q.x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) xh[j] = (__uint_as_float(wv[j]) - q.mean) * q.rstd;
    } else {
      unpack8(q.x0, xh);
#pragma unroll
      for (int j = 0; j < 8; ++j) xh[j] = (xh[j] - q.mean) * q.rstd;
    }
  };
  auto body = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    LnRaw<TX> cur, nxt;
    int row = blockIdx.x, it = 0;
    if (row < N_) fetch(row, cur);
    for (; row < N_; row += gridDim.x, ++it) {
      const int rn = row + gridDim.x;
      if (rn < N_) fetch(rn, nxt);                         // the next row is in flight while this one is reduced
      float g[8], xh[8];
      unpack8(cur.dy, g);
      xhat8(cur, xh);
      float ss[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dyn = in ? g[j] : 0.f;
        if constexpr (ACT != SMX_ACT_NONE) dyn *= act_grad_c<ACT>(xh[j] * gam[j] + bet[j]);
        const float xq = in ? xh[j] : 0.f;
        const float gg = dyn * gam[j];
        ss[0] += gg;
        ss[1] += gg * xq;
        dg[j] += dyn * xq;
        db[j] += dyn;
      }
      wg_sum<2>(ss, red[it & 1], lane, w);                 // (alternating buffers: ONE barrier per row)
      if (in) {
        const float m1 = ss[0] * invD, m2 = ss[1] * invD;
        float o[8];
        unpack8(cur.dy, g);                                // (unpacked again instead of kept: 16 registers less across the barrier)
        xhat8(cur, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dyn = g[j];
          if constexpr (ACT != SMX_ACT_NONE) dyn *= act_grad_c<ACT>(xh[j] * gam[j] + bet[j]);
          o[j] = cur.rstd * (dyn * gam[j] - m1 - xh[j] * m2);
        }
        if constexpr (!PRE) {
          if (R) {
            float rr[8];
            unpack8(cur.r, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rr[j];
          }
        }
        if constexpr (PRE) {
          float zz[8];
          unpack8(cur.z, zz);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= act_grad(zact, zz[j]);
        }
        st8_bf16(dX + (long)row * lddx + c, o);
      }
      cur = nxt;
    }
  };
  if constexpr (PRE) body(ActTag<SMX_ACT_NONE>{});           // (the PRE entry point takes a plain LayerNorm only: one instantiation, no spills)
  else dispatch_act(act, body);
  if (in) {                                                // ONE partial row pair per workgroup (fixed-order reduction downstream)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      partial[((long)blockIdx.x * 2) * D + c + j] = dg[j];
      partial[((long)blockIdx.x * 2 + 1) * D + c + j] = db[j];
    }
  }
}

// wide rows (2048 < D <= 4096, e.g. the (F,C) = 40x64 LayerNorm of the conv front-end): one WORKGROUP per row at a
// time, thread t owns columns t + 256*i; row statistics through an LDS reduction; same partial-row flush as above.
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ X,
                                                                 long ldx, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int act,
                                                                 const float* __restrict__ stats, const T* __restrict__ R,
                                                                 long ldr, T* __restrict__ dX, long lddx,
                                                                 float* __restrict__ partial, int N_, int D) {
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  float gam[CH], bet[CH], dg[CH], db[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = t + 256 * i;
    gam[i] = c < D ? gamma[c] : 0.f;
    bet[i] = (c < D && act != SMX_ACT_NONE) ? beta[c] : 0.f;
    dg[i] = db[i] = 0.f;
  }
  for (int row = blockIdx.x; row < N_; row += gridDim.x) {
    const float mean = stats[2 * (long)row], rstd = stats[2 * (long)row + 1];
    float g[CH], xh[CH], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = t + 256 * i;
      g[i] = xh[i] = 0.f;
      if (c < D) {
        const float xhat = (to_f32(X[(long)row * ldx + c]) - mean) * rstd;
        float dyn = to_f32(dY[(long)row * lddy + c]);
        if (act != SMX_ACT_NONE) dyn *= act_grad(act, xhat * gam[i] + bet[i]);
        g[i] = dyn * gam[i]; xh[i] = xhat;
        s1 += g[i]; s2 += g[i] * xhat;
        dg[i] += dyn * xhat; db[i] += dyn;
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __syncthreads();
    if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    const float m1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
    const float m2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = t + 256 * i;
      if (c < D) {
        float o = rstd * (g[i] - m1 - xh[i] * m2);
        if (R) o += to_f32(R[(long)row * ldr + c]);
        dX[(long)row * lddx + c] = from_f32<T>(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = t + 256 * i;
    if (c < D) { partial[((long)blockIdx.x * 2) * D + c] = dg[i]; partial[((long)blockIdx.x * 2 + 1) * D + c] = db[i]; }
  }
}

// the same with 4 consecutive columns per thread (8-byte accesses in bf16): thread t owns columns (t + 256 i) * 4 .. + 3.
// The 2-byte accesses of the kernel above cap it at the vector-memory issue rate (1.58 ms for the 2 GB of the front-end's
// (128128, 2560) LayerNorm backward = 1.3 TB/s).
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_wide4_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ X,
                                                                  long ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int act,
                                                                  const float* __restrict__ stats, const T* __restrict__ R,
                                                                  long ldr, T* __restrict__ dX, long lddx,
                                                                  float* __restrict__ partial, int N_, int D) {
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  float gam[CH][4], bet[CH][4], dg[CH][4], db[CH][4];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (t + 256 * i) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gam[i][q] = c < D ? gamma[c + q] : 0.f;
      bet[i][q] = (c < D && act != SMX_ACT_NONE) ? beta[c + q] : 0.f;
      dg[i][q] = db[i][q] = 0.f;
    }
  }
  for (int row = blockIdx.x; row < N_; row += gridDim.x) {
    const float mean = stats[2 * (long)row], rstd = stats[2 * (long)row + 1];
    float g[CH][4], xh[CH][4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (t + 256 * i) * 4;
      if (c < D) {
        load4<T>(X + (long)row * ldx + c, xh[i]);
        load4<T>(dY + (long)row * lddy + c, g[i]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) g[i][q] = xh[i][q] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xhat = (xh[i][q] - mean) * rstd;
        float dyn = g[i][q];
        if (act != SMX_ACT_NONE) dyn *= act_grad(act, xhat * gam[i][q] + bet[i][q]);
        const bool in = (t + 256 * i) * 4 < D;
        xh[i][q] = in ? xhat : 0.f;
        dyn = in ? dyn : 0.f;
        g[i][q] = dyn * gam[i][q];
        s1 += g[i][q]; s2 += g[i][q] * xh[i][q];
        dg[i][q] += dyn * xh[i][q]; db[i][q] += dyn;
      }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __syncthreads();
    if (lane == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    const float m1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
    const float m2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = (t + 256 * i) * 4;
      if (c < D) {
        float o[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
        if (R) load4<T>(R + (long)row * ldr + c, rr);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = rstd * (g[i][q] - m1 - xh[i][q] * m2) + rr[q];
        store4<T>(dX + (long)row * lddx + c, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = (t + 256 * i) * 4;
    if (c < D) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        partial[((long)blockIdx.x * 2) * D + c + q] = dg[i][q];
        partial[((long)blockIdx.x * 2 + 1) * D + c + q] = db[i][q];
      }
    }
  }
}

// dgamma[c] += sum_b partial[b][0][c]; dbeta[c] += sum_b partial[b][1][c]   (fixed order => bit-reproducible)
// block = 32 columns x 8 row groups; every thread sums nblocks/8 partial rows with 4 independent accumulators.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                              float* dgamma, float* dbeta) {
  __shared__ float red[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;                 // index into the concatenated [dgamma | dbeta] row of 2*D
  const bool ok = c < 2 * D;
  const int pass = ok ? c / D : 0, col = ok ? c % D : 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = ry;
  for (; b + 48 < nblocks; b += 64) {
    s0 += partial[((long)b * 2 + pass) * D + col];
    s1 += partial[((long)(b + 16) * 2 + pass) * D + col];
    s2 += partial[((long)(b + 32) * 2 + pass) * D + col];
    s3 += partial[((long)(b + 48) * 2 + pass) * D + col];
  }
  for (; b < nblocks; b += 16) s0 += partial[((long)b * 2 + pass) * D + col];
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && ok) {
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) tot += red[r][cx];
    float* dst = pass == 0 ? dgamma : dbeta;
    dst[col] += tot;
  }
}

// =================================================================================================
// dZ = alpha * dY * mask * act'(Z);  dbias[m] += colsum;  dgroup[n/div, m] += per-group colsum.
// VEC kernel: a lane owns VT<T>::N consecutive columns (16-byte accesses); LPR lanes span a row chunk of up to
// 64*N columns, the remaining 64/LPR lanes of a wave take further rows; 4 rows are in flight per lane.
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void act_mask_bwd_vec_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ Z,
                                                               long ldz, const uint8_t* __restrict__ mask, T* __restrict__ dZ,
                                                               long lddz, int N_, int M, int act, float alpha, float* dbias,
                                                               float* dgroup, long lddg, int gdiv, int RS, int LPR,
                                                               float* __restrict__ partial, uint32_t dthresh, float dscale,
                                                               uint64_t dseed_, const uint64_t* ep) {
  constexpr int N = VT<T>::N;
  const uint64_t dseed = dthresh ? epoch_seed(dseed_, ep) : 0;
  __shared__ float red[256][N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lc = lane % LPR, lr = lane / LPR;          // column chunk / row slot inside the wave
  const int RPW = 64 / LPR;                            // rows per wave instruction
  const int col = (blockIdx.x * LPR + lc) * N;
  const bool cok = col < M;
  const int colc = cok ? col : 0;
  float bsum[N], gsum[N];
#pragma unroll
  for (int q = 0; q < N; ++q) bsum[q] = gsum[q] = 0.f;
  int cur_g = -1;
  const int rstep = 4 * RPW;                           // rows advanced per pass of the block's 4 waves
  dispatch_act(Z ? act : SMX_ACT_NONE, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    // a workgroup strides over row strips: gridDim.y (<= ACT_BWD_YMAX) partial rows for the bias reduction however long N is
    for (int r0 = blockIdx.y * RS; r0 < N_; r0 += gridDim.y * RS) {
    const int r1 = min(N_, r0 + RS);
    for (int nb = r0 + w * RPW + lr; nb < r1; nb += 4 * rstep) {
      float fdy[4][N], fz[4][N];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = min(nb + u * rstep, N_ - 1);
        loadv<T, true>(dY + (long)n * lddy + colc, N, fdy[u]);
        if (ACT != SMX_ACT_NONE) loadv<T, true>(Z + (long)n * ldz + colc, N, fz[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = nb + u * rstep;
        if (n >= r1 || !cok) continue;
        const float mk = (mask ? (mask[n] ? 1.f : 0.f) : 1.f) * alpha;
        float o[N];
#pragma unroll
        for (int q = 0; q < N; ++q) o[q] = fdy[u][q] * mk * (ACT != SMX_ACT_NONE ? act_grad_c<ACT>(fz[u][q]) : 1.f);
        if (dthresh) dropout_apply_any<N>(o, dseed, (uint64_t)n * M + col, dthresh, dscale);
        if (dZ) storev<T, true>(dZ + (long)n * lddz + col, N, o);
#pragma unroll
        for (int q = 0; q < N; ++q) bsum[q] += o[q];
        if (dgroup) {
          const int g = n / gdiv;
          if (g != cur_g) {
            if (cur_g >= 0) { for (int q = 0; q < N; ++q) atomicAdd(dgroup + (long)cur_g * lddg + col + q, gsum[q]); }
            cur_g = g;
#pragma unroll
            for (int q = 0; q < N; ++q) gsum[q] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < N; ++q) gsum[q] += o[q];
        }
      }
    }
    }
  });
  if (dgroup && cur_g >= 0 && cok) { for (int q = 0; q < N; ++q) atomicAdd(dgroup + (long)cur_g * lddg + col + q, gsum[q]); }
  if (dbias) {                                         // fold the 256 threads that share a column chunk
#pragma unroll
    for (int q = 0; q < N; ++q) red[threadIdx.x][q] = bsum[q];
    __syncthreads();
    if (threadIdx.x < LPR && cok) {
      float tot[N];
#pragma unroll
      for (int q = 0; q < N; ++q) tot[q] = 0.f;
      for (int k = threadIdx.x; k < 256; k += LPR) {
#pragma unroll
        for (int q = 0; q < N; ++q) tot[q] += red[k][q];
      }
      // one partial row per row strip (no atomics: ~100 ns each when thousands of blocks hit the same address)
#pragma unroll
      for (int q = 0; q < N; ++q) partial[(long)blockIdx.y * M + col + q] = tot[q];
    }
  }
}

// dst[c] += sum_b partial[b][c]  (fixed order).  block = 32 columns x 8 row groups
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, int nrows, int W, float* dst) {
  __shared__ float red[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  const bool ok = c < W;
  const int col = ok ? c : 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = ry;
  for (; b + 48 < nrows; b += 64) {
    s0 += partial[(long)b * W + col]; s1 += partial[(long)(b + 16) * W + col];
    s2 += partial[(long)(b + 32) * W + col]; s3 += partial[(long)(b + 48) * W + col];
  }
  for (; b < nrows; b += 16) s0 += partial[(long)b * W + col];
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && ok) {
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) tot += red[r][cx];
    dst[col] += tot;
  }
}

void launch_colsum_partials(const float* partial, int nrows, int W, float* dst, hipStream_t stream) {
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((W + 15) / 16), dim3(256), 0, stream, partial, nrows, W, dst);
}

// generic (ragged / unaligned) variant: 4 columns per thread, scalar accesses
template <typename T>
__global__ __launch_bounds__(256) void act_mask_bwd_kernel(const T* dY, long lddy, const T* Z, long ldz,
                                                           const uint8_t* mask, T* dZ, long lddz, int N_, int M, int act,
                                                           float alpha, float* dbias, float* dgroup, long lddg, int gdiv,
                                                           int RS, uint32_t dthresh, float dscale, uint64_t dseed_,
                                                           const uint64_t* ep) {
  const uint64_t dseed = dthresh ? epoch_seed(dseed_, ep) : 0;
  __shared__ float red[3][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cx) * 4;
  const int nvalid = M - col;
  const int r0 = blockIdx.y * RS, r1 = min(N_, r0 + RS);
  float bsum[4] = {0.f, 0.f, 0.f, 0.f}, gsum[4] = {0.f, 0.f, 0.f, 0.f};
  int cur_g = -1;
  if (nvalid > 0) {
    for (int n = r0 + ry; n < r1; n += 4) {
      float fdy[4], fz[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
      for (int q = 0; q < 4; ++q) fdy[q] = q < nvalid ? to_f32(dY[(long)n * lddy + col + q]) : 0.f;
      if (Z) { for (int q = 0; q < 4; ++q) fz[q] = q < nvalid ? to_f32(Z[(long)n * ldz + col + q]) : 0.f; }
      const float mk = (mask ? (mask[n] ? 1.f : 0.f) : 1.f) * alpha;
      for (int q = 0; q < 4; ++q) o[q] = fdy[q] * mk * (Z ? act_grad(act, fz[q]) : 1.f);
      if (dthresh) dropout_apply_any<4>(o, dseed, (uint64_t)n * M + col, dthresh, dscale);
      if (dZ) { for (int q = 0; q < 4; ++q) if (q < nvalid) dZ[(long)n * lddz + col + q] = from_f32<T>(o[q]); }
      for (int q = 0; q < 4; ++q) bsum[q] += o[q];
      if (dgroup) {
        int g = n / gdiv;
        if (g != cur_g) {
          if (cur_g >= 0) { for (int q = 0; q < 4; ++q) if (q < nvalid) atomicAdd(dgroup + (long)cur_g * lddg + col + q, gsum[q]); }
          cur_g = g;
          for (int q = 0; q < 4; ++q) gsum[q] = 0.f;
        }
        for (int q = 0; q < 4; ++q) gsum[q] += o[q];
      }
    }
    if (dgroup && cur_g >= 0) { for (int q = 0; q < 4; ++q) if (q < nvalid) atomicAdd(dgroup + (long)cur_g * lddg + col + q, gsum[q]); }
  }
  if (dbias) {
    if (ry > 0) { for (int q = 0; q < 4; ++q) red[ry - 1][cx][q] = bsum[q]; }
    __syncthreads();
    if (ry == 0 && nvalid > 0) {
      for (int q = 0; q < 4; ++q)
        if (q < nvalid) atomicAdd(dbias + col + q, ((bsum[q] + red[0][cx][q]) + red[1][cx][q]) + red[2][cx][q]);
    }
  }
}

// y = a*x + b*y0
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void axpby_kernel(float a, const T* X, long ldx, float b, const T* Y0, long ldy0, T* Y,
                                                    long ldy, int N_, int D) {
  const int cv = (D + 3) / 4;
  long total = (long)N_ * cv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int n = (int)(i / cv), col = (int)(i % cv) * 4, nvalid = D - col;
    float fx[4], fy[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
    if (VEC && nvalid >= 4) load4<T>(X + (long)n * ldx + col, fx);
    else { for (int q = 0; q < 4; ++q) fx[q] = q < nvalid ? to_f32(X[(long)n * ldx + col + q]) : 0.f; }
    if (Y0) {
      if (VEC && nvalid >= 4) load4<T>(Y0 + (long)n * ldy0 + col, fy);
      else { for (int q = 0; q < 4; ++q) fy[q] = q < nvalid ? to_f32(Y0[(long)n * ldy0 + col + q]) : 0.f; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = a * fx[q] + b * fy[q];
    if (VEC && nvalid >= 4) store4<T>(Y + (long)n * ldy + col, o);
    else { for (int q = 0; q < 4; ++q) if (q < nvalid) Y[(long)n * ldy + col + q] = from_f32<T>(o[q]); }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* src, T* dst, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = from_f32<T>(src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* src, float* dst, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = to_f32(src[i]);
}

// fused AdamW (torch.optim.AdamW semantics: p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps))
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, uint16_t* shadow, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float gscale, const float* gscale_dev,
                                                    const uint64_t* step_dev) {
  // a clip factor of exactly 0 is smx_clip_factor's "non-finite gradient norm" verdict: skip the whole update
  // (no weight decay, no moment update, shadows untouched) like SpeechBrain's check_gradients does
  if (gscale_dev && gscale_dev[0] == 0.f) return;
  const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
  if (step_dev) {                                        // bias correction from the device step counter (graph replay)
    const float t = (float)step_dev[0];
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i] * gs, pi = p[i], mi = m[i], vi = v[i];
    pi *= 1.f - lr * wd;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = (uint16_t)f32_to_bf16_bits(pi);
  }
}

// Global gradient norm in a FIXED order (no atomics): per-block partial sums, then one block folds them.  Data-parallel
// ranks hold bit-identical all-reduced gradients; an order-dependent norm would give them clip factors that differ in the
// last bits and let their weights drift apart.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int np, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void clip_factor_kernel(const float* sumsq, float max_norm, float inv_scale, float* out) {
  const float ss = sumsq[0];
  if (!(ss >= 0.f && ss < __builtin_inff())) {           // NaN or Inf gradient norm: factor 0 = "skip this step"
    out[0] = 0.f;
    out[1] += 1.f;                                        // skipped-step counter
    return;
  }
  float nrm = sqrtf(ss) * inv_scale;
  float f = max_norm / (nrm + 1e-6f);
  out[0] = f < 1.f ? f : 1.f;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* X, long ldx, T* Y, long ldy, int N_, int D, uint32_t thresh,
                                                      float scale, uint64_t seed_, const uint64_t* ep) {
  const uint64_t seed = epoch_seed(seed_, ep);
  const int cv = (D + 3) / 4;
  const long total = (long)N_ * cv;
  const bool vec = (D & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ((uintptr_t)X % (4 * sizeof(T))) == 0 &&
                   ((uintptr_t)Y % (4 * sizeof(T))) == 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / cv), col = (int)(i % cv) * 4, nvalid = D - col;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) load4<T>(X + (long)n * ldx + col, f);
    else { for (int q = 0; q < 4; ++q) if (q < nvalid) f[q] = to_f32(X[(long)n * ldx + col + q]); }
    dropout_apply_any<4>(f, seed, (uint64_t)n * D + col, thresh, scale);
    if (vec) store4<T>(Y + (long)n * ldy + col, f);
    else { for (int q = 0; q < 4; ++q) if (q < nvalid) Y[(long)n * ldy + col + q] = from_f32<T>(f[q]); }
  }
}
// Y[n, :] += table[n % R, :]   (fp32 table; the abs-sine positional encoding added after the input dropout)
template <typename T>
__global__ __launch_bounds__(256) void add_rowtable_kernel(T* Y, long ldy, const float* table, int R, int N_, int D) {
  const long total = (long)N_ * D;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / D), c = (int)(i % D);
    Y[(long)n * ldy + c] = from_f32<T>(to_f32(Y[(long)n * ldy + c]) + table[(long)(n % R) * D + c]);
  }
}

static inline int grid1d(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace smx

using namespace smx;
#define STREAM reinterpret_cast<hipStream_t>(stream)

extern "C" size_t smx_masked_mean_workspace(int B, int T, int D) {
  int DC, TS, TR;
  mm_plan(B, T, D, 4, DC, TS, TR);  // fp32 plan has the most splits; bf16 needs no more
  int DC2, TS2, TR2;
  mm_plan(B, T, D, 8, DC2, TS2, TR2);
  int ts = TS > TS2 ? TS : TS2;
  return (size_t)B * ts * (D + 1) * sizeof(float);
}

extern "C" int smx_masked_mean_fwd(int dtype, const void* S, int64_t lds, const uint8_t* mask, float* out,
                                   float* inv_count, int B, int T, int D, int scale_by_count, void* workspace,
                                   void* stream) {
  SMX_REQUIRE(S && out && workspace, "smx_masked_mean_fwd: null pointer");
  SMX_REQUIRE(B > 0 && T > 0 && D > 0, "smx_masked_mean_fwd: bad sizes");
  const int nvec = dtype == SMX_BF16 ? 8 : 4;
  int DC, TS, TR;
  mm_plan(B, T, D, nvec, DC, TS, TR);
  dim3 g1(DC, TS, B);
  float* partial = reinterpret_cast<float*>(workspace);
  float* pcount = partial + (size_t)B * TS * D;
  const bool vec = vec_ok(S, lds, D, nvec, dtype == SMX_BF16 ? 2 : 4);
  if (dtype == SMX_BF16) {
    if (vec) hipLaunchKernelGGL((masked_sum_stage1<bf16_t, true>), g1, dim3(256), 0, STREAM, (const bf16_t*)S, lds, mask, partial, pcount, T, D, TR, TS);
    else hipLaunchKernelGGL((masked_sum_stage1<bf16_t, false>), g1, dim3(256), 0, STREAM, (const bf16_t*)S, lds, mask, partial, pcount, T, D, TR, TS);
  } else if (dtype == SMX_F32) {
    if (vec) hipLaunchKernelGGL((masked_sum_stage1<float, true>), g1, dim3(256), 0, STREAM, (const float*)S, lds, mask, partial, pcount, T, D, TR, TS);
    else hipLaunchKernelGGL((masked_sum_stage1<float, false>), g1, dim3(256), 0, STREAM, (const float*)S, lds, mask, partial, pcount, T, D, TR, TS);
  } else return fail(SMX_EINVAL, "smx_masked_mean_fwd: bad dtype");
  hipLaunchKernelGGL(masked_sum_stage2, dim3((D + 255) / 256, B), dim3(256), 0, STREAM, partial, pcount, out, inv_count, T, D, TS, scale_by_count);
  return check_launch("smx_masked_mean_fwd");
}

#ifndef SMX_BCAST_RPB
#define SMX_BCAST_RPB 64       // rows per workgroup of the broadcast kernels that also read (act / mask backward)
#endif
#ifndef SMX_BCAST_RPB_ST
#define SMX_BCAST_RPB_ST 128   // ... of the store-only ones (repeat, repeat + dropout)
#endif
static int bcast_impl(int dtype, const float* g, const float* inv_count, void* dS, int64_t ldds, int B, int T, int D,
                      float drop_p, uint64_t drop_seed, const uint64_t* epoch, const void* Z, int64_t ldz, const uint8_t* mask, int act, void* stream,
                      const char* what) {
  const uint32_t dthresh = (uint32_t)((double)drop_p * 4294967296.0);
  const float dscale = 1.f / (1.f - drop_p);
  const int nvec = dtype == SMX_BF16 ? 8 : 4;
  int LPR = 1;
  while (LPR < 64 && LPR * nvec < D) LPR <<= 1;           // lanes per row (power of two)
  const int DC = (D + LPR * nvec - 1) / (LPR * nvec);
  int RPB = (Z || mask) ? SMX_BCAST_RPB : SMX_BCAST_RPB_ST;
  while (RPB > 8 && (long)DC * ((T + RPB - 1) / RPB) * B < 512) RPB >>= 1;   // small batches: more, shorter workgroups (the recipe's 10 x 375 frames ran on 30)
  dim3 grid(DC, (T + RPB - 1) / RPB, B);
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  const bool vec = vec_ok(dS, ldds, D, nvec, es) && (Z == nullptr || vec_ok(Z, ldz, D, nvec, es));
  if (dtype == SMX_BF16) {
    if (vec) hipLaunchKernelGGL((bcast_rows_kernel<bf16_t, true>), grid, dim3(256), 0, STREAM, g, inv_count, (bf16_t*)dS, ldds, T, D, RPB, dthresh, dscale, drop_seed, epoch, (const bf16_t*)Z, ldz, mask, act, LPR);
    else hipLaunchKernelGGL((bcast_rows_kernel<bf16_t, false>), grid, dim3(256), 0, STREAM, g, inv_count, (bf16_t*)dS, ldds, T, D, RPB, dthresh, dscale, drop_seed, epoch, (const bf16_t*)Z, ldz, mask, act, LPR);
  } else {
    if (vec) hipLaunchKernelGGL((bcast_rows_kernel<float, true>), grid, dim3(256), 0, STREAM, g, inv_count, (float*)dS, ldds, T, D, RPB, dthresh, dscale, drop_seed, epoch, (const float*)Z, ldz, mask, act, LPR);
    else hipLaunchKernelGGL((bcast_rows_kernel<float, false>), grid, dim3(256), 0, STREAM, g, inv_count, (float*)dS, ldds, T, D, RPB, dthresh, dscale, drop_seed, epoch, (const float*)Z, ldz, mask, act, LPR);
  }
  return check_launch(what);
}

extern "C" int smx_masked_mean_bwd(int dtype, const float* g, const float* inv_count, void* dS, int64_t ldds, int B,
                                   int T, int D, float drop_p, uint64_t drop_seed, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(g && dS, "smx_masked_mean_bwd: null pointer");
  SMX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "smx_masked_mean_bwd: 0 <= drop_p < 1");
  return bcast_impl(dtype, g, inv_count, dS, ldds, B, T, D, drop_p, drop_seed, epoch, nullptr, 0, nullptr, SMX_ACT_NONE, stream,
                    "smx_masked_mean_bwd");
}

extern "C" int smx_masked_mean_bwd_act(int dtype, const float* g, const float* inv_count, void* dS, int64_t ldds,
                                       const void* Z, int64_t ldz, const uint8_t* row_mask, int act, int B, int T, int D,
                                       void* stream) {
  SMX_REQUIRE(g && dS && (Z || row_mask), "smx_masked_mean_bwd_act: null pointer");
  return bcast_impl(dtype, g, inv_count, dS, ldds, B, T, D, 0.f, 0, nullptr, Z, ldz, row_mask, Z ? act : SMX_ACT_NONE, stream,
                    "smx_masked_mean_bwd_act");
}

// =================================================================================================
// Round 6, small batches: the masked mean AND its broadcast in ONE launch.  The per-utterance summary is three launches on the big
// path (split-T partial sums, their fixed-order fold, the broadcast back over t) because 64 000 frames want thousands of
// workgroups; the recipe's 10 x 375 frames or one utterance want neither - every launch of a replayed step costs >= 4.5 us
// whatever it does (profiles/r06_seq_c2a_recipe_batch_v0_baseline.txt: 6.6 + 4.8 + 6.3 us for 3.8 MB in, 3.8 MB out).
// One workgroup owns (utterance b, a group of 8 lanes x VT::N columns): its 32 row slots walk the T frames (4 rows in flight each),
// fold through LDS in a FIXED order (bit-reproducible, no atomics), then the same threads write the broadcast with the options of
// bcast_rows_kernel: inverted dropout (forward `repeat`, summary_mixing.py:222,237-239,267) or the act / mask backward of the
// projection that produced the summary columns (backward of summary_mixing.py:210,257).
// =================================================================================================
// CL lanes x VT::N columns per row and workgroup, RS = 256 / CL row slots (chosen at launch: narrower groups = more workgroups
// and more rows in flight for the 10-utterance recipe batch or a single utterance - the first version, 8 lanes and 4 rows in flight
// on 80 workgroups, took 10-15 us: three dependent round trips per slot).
#ifndef SMX_PB_UB
#define SMX_PB_UB 4      // rows in flight of the backward's write phase
#endif
template <typename T, bool VEC, int CL>
__global__ __launch_bounds__(256) void pool_bcast_kernel(const T* __restrict__ S, long lds, const uint8_t* __restrict__ mask_in,
                                                         float* mean_out, const float* __restrict__ inv_in, float* inv_out,
                                                         T* dS, long ldds, int T_, int D, int scale, uint32_t dthresh, float dscale,
                                                         uint64_t dseed_, const uint64_t* ep, const T* __restrict__ Z, long ldz,
                                                         const uint8_t* __restrict__ mask_out, int act) {
  constexpr int N = VT<T>::N, RS = 256 / CL, U = 8, G = 8, RG = RS / G;   // columns per lane, row slots, rows in flight, fold groups
  __shared__ float red[RS][CL * N + 1];
  __shared__ float cred[RS];
  const uint64_t dseed = dthresh ? epoch_seed(dseed_, ep) : 0;
  const int cl = threadIdx.x % CL, rs = threadIdx.x / CL, b = blockIdx.y;
  const int col = (blockIdx.x * CL + cl) * N;
  const int nvalid = D - col;                              // (<= 0: an idle lane of the last column group - it still joins the barriers)
  const T* base = S + (long)b * T_ * lds + (nvalid > 0 ? col : 0);
  const uint8_t* mrow = mask_in ? mask_in + (long)b * T_ : nullptr;
  float acc[N], cnt = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.f;
  const int tlast = T_ - 1;
  for (int t = rs; t < T_; t += U * RS) {                  // U rows in flight per slot; rows past the end: the last row, weight 0
    float f[U][N], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) loadv<T, VEC>(base + (long)min(t + RS * u, tlast) * lds, nvalid > 0 ? nvalid : 1, f[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tt = t + RS * u;
      m[u] = tt < T_ ? (mrow ? (mrow[tt] ? 1.f : 0.f) : 1.f) : 0.f;
      cnt += m[u];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc[i] += ((f[0][i] * m[0] + f[1][i] * m[1]) + (f[2][i] * m[2] + f[3][i] * m[3])) +
                ((f[4][i] * m[4] + f[5][i] * m[5]) + (f[6][i] * m[6] + f[7][i] * m[7]));
  }
#pragma unroll
  for (int i = 0; i < N; ++i) red[rs][cl * N + i] = acc[i];
  if (cl == 0) cred[rs] = cnt;
  __syncthreads();
  // fixed-order fold in two levels: slot g of the first G folds slots g * RG .. g * RG + RG - 1, then everybody folds the G results
  if (rs < G) {
    float c0 = 0.f, w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = 0.f;
    for (int k = rs * RG; k < rs * RG + RG; ++k) {
      c0 += cred[k];
#pragma unroll
      for (int i = 0; i < N; ++i) w[i] += red[k][cl * N + i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // (lgkmcnt(0): the reads above are done before the slot is overwritten)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) red[rs * RG][cl * N + i] = w[i];
    if (cl == 0) cred[rs * RG] = c0;
  }
  __syncthreads();
  float v[N], c = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    c += cred[g * RG];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += red[g * RG][cl * N + i];
  }
  const float inv = 1.f / c;   // zero valid frames -> inf, 0 * inf = NaN, as the reference (summary_mixing.py:264-266)
  if (inv_out && blockIdx.x == 0 && threadIdx.x == 0) inv_out[b] = inv;
  const float sc = (scale ? inv : 1.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] *= sc;
  if (nvalid <= 0) return;
  if (mean_out && rs == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < nvalid) mean_out[(long)b * D + col + i] = v[i];
  }
  if (!dS) return;
  if (inv_in) {
    const float si = inv_in[b];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] *= si;
  }
  if (Z || mask_out) {                                     // backward through act(.) * mask of the producing projection
    dispatch_act(Z ? act : SMX_ACT_NONE, [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      constexpr int UB = SMX_PB_UB;                         // rows in flight
      for (int t = rs; t < T_; t += UB * RS) {
        float z[UB][N], mk[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const long n = (long)b * T_ + min(t + u * RS, tlast);
          mk[u] = mask_out ? (mask_out[n] ? 1.f : 0.f) : 1.f;
          if (ACT != SMX_ACT_NONE) loadv<T, VEC>(Z + n * ldz + col, nvalid, z[u]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (t + u * RS >= T_) break;
          float o[N];
#pragma unroll
          for (int i = 0; i < N; ++i) o[i] = ACT != SMX_ACT_NONE ? v[i] * mk[u] * act_grad_c<ACT>(z[u][i]) : v[i] * mk[u];
          storev<T, VEC>(dS + ((long)b * T_ + t + u * RS) * ldds + col, nvalid, o);
        }
      }
    });
  } else if (dthresh == 0) {
    for (int t = rs; t < T_; t += RS) storev<T, VEC>(dS + ((long)b * T_ + t) * ldds + col, nvalid, v);
  } else {                                                 // fused inverted dropout, mask = f(seed, row * D + col)
    for (int t = rs; t < T_; t += RS) {
      const uint64_t ix = ((uint64_t)b * T_ + t) * (uint64_t)D + col;
      float o[N];
#pragma unroll
      for (int i = 0; i < N; ++i) o[i] = v[i];
      dropout_apply_any<N>(o, dseed, ix, dthresh, dscale);
      storev<T, VEC>(dS + ((long)b * T_ + t) * ldds + col, nvalid, o);
    }
  }
}

extern "C" int smx_pool_bcast_ok(int B, int T, int D) {
  // a workgroup per (utterance, 64 bf16 / 32 fp32 columns) walks all T frames: small batches - and big ones of ordinary utterances whose
  // (utterance, column group) pairs alone fill the chip (same-box A/B of the steps with it on: C2b B = 128 18.02 -> 17.90 ms, B = 64
  // 12.00 -> 11.88, C2a 45.36 -> 45.16, C4 38.51 -> 38.00).  Long utterances (config 5: T = 30 000) keep the split-T kernels, whose
  // thousands of workgroups stream at 0.72 of the HBM roof.  SMX_POOL_FUSE_MAX_ROWS: the frame count up to which few pairs are enough.
  if (!(B >= 1 && T >= 1 && D >= 1 && T <= 4096)) return 0;
  return (long)B * T <= (long)cfg().pool_fuse_max_rows || (long)B * ((D + 63) / 64) >= 256;
}

extern "C" int smx_pool_bcast(int dtype, const void* S, int64_t lds, const uint8_t* mask_in, float* mean_out, const float* inv_in,
                              float* inv_out, void* dS, int64_t ldds, int B, int T, int D, int scale_by_count, float drop_p,
                              uint64_t drop_seed, const uint64_t* epoch, const void* Z, int64_t ldz, const uint8_t* mask_out, int act,
                              void* stream) {
  SMX_REQUIRE(S && (mean_out || dS), "smx_pool_bcast: null pointer");
  SMX_REQUIRE(B > 0 && T > 0 && D > 0, "smx_pool_bcast: bad sizes");
  SMX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "smx_pool_bcast: 0 <= drop_p < 1");
  SMX_REQUIRE(!(drop_p > 0.f && (Z || mask_out)), "smx_pool_bcast: dropout (forward) and the act / mask backward exclude each other");
  if (!smx_pool_bcast_ok(B, T, D)) return fail(SMX_EUNSUPPORTED, "smx_pool_bcast: built for small batches (smx_pool_bcast_ok); use smx_masked_mean_fwd + smx_masked_mean_bwd");
  const uint32_t dthresh = (uint32_t)((double)drop_p * 4294967296.0);
  const float dscale = 1.f / (1.f - drop_p);
  const int nvec = dtype == SMX_BF16 ? 8 : 4;
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  const bool vec = vec_ok(S, lds, D, nvec, es) && (dS == nullptr || vec_ok(dS, ldds, D, nvec, es)) && (Z == nullptr || vec_ok(Z, ldz, D, nvec, es));
  const int a = Z ? act : SMX_ACT_NONE;
  // lanes per row: the widest group (whole 128-byte row segments) that still gives ~half a workgroup per CU
  int CL = 8;
  while (CL > 2 && (long)B * ((D + CL * nvec - 1) / (CL * nvec)) < 128) CL >>= 1;
  dim3 grid((D + CL * nvec - 1) / (CL * nvec), B);
#define SMX_PB_LAUNCH(TT, VV, CC) hipLaunchKernelGGL((pool_bcast_kernel<TT, VV, CC>), grid, dim3(256), 0, STREAM, (const TT*)S, lds, mask_in, mean_out, inv_in, inv_out, (TT*)dS, ldds, T, D, scale_by_count, dthresh, dscale, drop_seed, epoch, (const TT*)Z, ldz, mask_out, a)
#define SMX_PB_CL(TT, VV) do { if (CL == 8) SMX_PB_LAUNCH(TT, VV, 8); else if (CL == 4) SMX_PB_LAUNCH(TT, VV, 4); else SMX_PB_LAUNCH(TT, VV, 2); } while (0)
  if (dtype == SMX_BF16) {
    if (vec) SMX_PB_CL(bf16_t, true); else SMX_PB_CL(bf16_t, false);
  } else if (dtype == SMX_F32) {
    if (vec) SMX_PB_CL(float, true); else SMX_PB_CL(float, false);
  } else return fail(SMX_EINVAL, "smx_pool_bcast: bad dtype");
#undef SMX_PB_CL
#undef SMX_PB_LAUNCH
  return check_launch("smx_pool_bcast");
}

extern "C" size_t smx_chunk_mean_workspace(int B, int T, int D, int chunk) {
  int NC = (T + chunk - 1) / chunk;
  return (size_t)B * NC * D * sizeof(float);
}

// phase: 3 = the whole operator (one shard = the whole sequence); 1 = chunk sums (+ running sums) only - the shard's exchange
// values are then rows of the workspace; 2 = the window combine, with the other shards' contributions in `carry`
template <typename T>
static int chunk_mean_impl(const void* X, int64_t ldx, void* out, int64_t ldo, int B, int T_, int D, int chunk, int left,
                           int reverse, void* ws, hipStream_t s, int phase = 3, int c_off = 0, const float* carry = nullptr,
                           int carry_c0 = 0, int carry_n = 0) {
  const int nvec = VT<T>::N;
  const int NC = (T_ + chunk - 1) / chunk, DC = (D + 64 * nvec - 1) / (64 * nvec);
  dim3 grid(DC, NC, B);
  float* csum = reinterpret_cast<float*>(ws);
  const bool v1 = vec_ok(X, ldx, D, nvec, sizeof(T)), v2 = vec_ok(out, ldo, D, nvec, sizeof(T));
  if (phase & 1) {
    if (v1) hipLaunchKernelGGL((chunk_sum_kernel<T, true>), grid, dim3(256), 0, s, (const T*)X, ldx, csum, T_, D, chunk, NC, left, reverse, c_off);
    else hipLaunchKernelGGL((chunk_sum_kernel<T, false>), grid, dim3(256), 0, s, (const T*)X, ldx, csum, T_, D, chunk, NC, left, reverse, c_off);
    if (left < 0) hipLaunchKernelGGL(chunk_prefix_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)B), dim3(256), 0, s, csum, D, NC, reverse);
  }
  if (phase & 2) {
    if (v2) hipLaunchKernelGGL((chunk_window_kernel<T, true>), grid, dim3(256), 0, s, csum, (T*)out, ldo, T_, D, chunk, NC, left, reverse, c_off, carry, carry_c0, carry_n);
    else hipLaunchKernelGGL((chunk_window_kernel<T, false>), grid, dim3(256), 0, s, csum, (T*)out, ldo, T_, D, chunk, NC, left, reverse, c_off, carry, carry_c0, carry_n);
  }
  return check_launch("smx_chunk_mean");
}

extern "C" int smx_chunk_mean_fwd(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D,
                                  int chunk, int left, void* workspace, void* stream) {
  SMX_REQUIRE(S && out && workspace && chunk > 0, "smx_chunk_mean_fwd: bad arguments");
  if (dtype == SMX_BF16) return chunk_mean_impl<bf16_t>(S, lds, out, ldo, B, T, D, chunk, left, 0, workspace, STREAM);
  return chunk_mean_impl<float>(S, lds, out, ldo, B, T, D, chunk, left, 0, workspace, STREAM);
}
extern "C" int smx_chunk_mean_bwd(int dtype, const void* dOut, int64_t ldo, void* dS, int64_t lds, int B, int T, int D,
                                  int chunk, int left, void* workspace, void* stream) {
  SMX_REQUIRE(dOut && dS && workspace && chunk > 0, "smx_chunk_mean_bwd: bad arguments");
  if (dtype == SMX_BF16) return chunk_mean_impl<bf16_t>(dOut, ldo, dS, lds, B, T, D, chunk, left, 1, workspace, STREAM);
  return chunk_mean_impl<float>(dOut, ldo, dS, lds, B, T, D, chunk, left, 1, workspace, STREAM);
}

// Sequence-parallel shards of whole chunks (sequence_parallel.py; summary_mixing.py:224-235 with the mask of TransformerASR.py:85-110):
// the shard holds chunks [c_off, c_off + T / chunk) of the sequence.  phase 1: chunk sums into the workspace (B, T / chunk, D) float32
// - reverse: scaled by 1 / window length of the GLOBAL chunk; left < 0: turned into running sums - whose rows are what the shards
// exchange (the last `left` rows / the last running sum forward, the first ones in reverse).  phase 2: the window combine; chunk c
// adds carry[b][c - carry_c0] (float32 (B, carry_n, D); carry_n == 0: one (B, D) row for every chunk), forward divides by the
// GLOBAL window length.  No elementwise pass over the activations happens outside these kernels.
extern "C" int smx_chunk_mean_sharded(int dtype, const void* X, int64_t ldx, void* out, int64_t ldo, int B, int T, int D, int chunk,
                                      int left, int reverse, int c_off, int phase, const float* carry, int carry_c0, int carry_n,
                                      void* workspace, void* stream) {
  SMX_REQUIRE(workspace && chunk > 0 && T % chunk == 0 && c_off >= 0 && (phase == 1 || phase == 2) && carry_n >= 0 && carry_c0 >= 0,
              "smx_chunk_mean_sharded: shards hold whole chunks; phase 1 or 2");
  SMX_REQUIRE(phase == 1 ? X != nullptr : out != nullptr, "smx_chunk_mean_sharded: null pointer");
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  if (dtype == SMX_BF16) return chunk_mean_impl<bf16_t>(X, ldx, out, ldo, B, T, D, chunk, left, reverse, workspace, STREAM, phase, c_off, carry, carry_c0, carry_n);
  return chunk_mean_impl<float>(X, ldx, out, ldo, B, T, D, chunk, left, reverse, workspace, STREAM, phase, c_off, carry, carry_c0, carry_n);
}

extern "C" size_t smx_expdecay_mean_workspace(int B, int T, int D) {
  return (size_t)B * ((T + ED_CH - 1) / ED_CH) * 2 * D * sizeof(float);
}

// phase 3: the whole operator; 1: chunk states + the shard's LEAVING states into `ends` (2, B, D); 2: carries (from the ENTERING states
// in `ends`) + both scans out (the workspace of phase 1 must still hold the chunk states)
template <typename T>
static int expdecay_impl(const void* X, int64_t ldx, void* Y, int64_t ldy, int B, int T_, int D, float gamma, int mode,
                         void* ws, hipStream_t s, int phase = 3, int t_off = 0, int T_glob = 0, float* ends = nullptr) {
  const int NC = (T_ + ED_CH - 1) / ED_CH;
  if (T_glob <= 0) T_glob = T_;
  dim3 grid((D + 255) / 256, (NC + 3) / 4, B);
  const dim3 gc((unsigned)(((long)B * D + 255) / 256));
  float* w = reinterpret_cast<float*>(ws);
  if (phase & 1) hipLaunchKernelGGL((expdecay_chunk_kernel<T>), grid, dim3(256), 0, s, (const T*)X, ldx, w, T_, D, NC, gamma, mode, t_off, T_glob);
  if (phase == 1) hipLaunchKernelGGL(expdecay_carry_kernel, gc, dim3(256), 0, s, w, D, NC, B, gamma, (const float*)nullptr, (const float*)nullptr, ends, T_);
  if (phase & 2) {
    const float* fin = phase == 2 ? ends : nullptr;
    hipLaunchKernelGGL(expdecay_carry_kernel, gc, dim3(256), 0, s, w, D, NC, B, gamma, fin, fin ? fin + (long)B * D : nullptr, (float*)nullptr, T_);
    hipLaunchKernelGGL((expdecay_apply_kernel<T>), grid, dim3(256), 0, s, (const T*)X, ldx, w, (T*)Y, ldy, T_, D, NC, gamma, mode, t_off, T_glob);
  }
  return check_launch("smx_expdecay_mean");
}

static int expdecay_check(const void* X, int64_t ldx, const void* Y, int64_t ldy, int D, float gamma, const void* ws, size_t es) {
  SMX_REQUIRE(X && Y && ws, "smx_expdecay_mean: null pointer");
  SMX_REQUIRE(gamma > 0.f && gamma < 1.f, "smx_expdecay_mean: 0 < decay < 1 (got %f)", (double)gamma);
  SMX_REQUIRE(D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(X) % (4 * es)) == 0 &&
                  (reinterpret_cast<uintptr_t>(Y) % (4 * es)) == 0 && aligned16(ws),
              "smx_expdecay_mean: needs D, leading dimensions multiples of 4 and aligned pointers");
  return SMX_OK;
}

extern "C" int smx_expdecay_mean_fwd(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D,
                                     float decay, void* workspace, void* stream) {
  int rc = expdecay_check(S, lds, out, ldo, D, decay, workspace, dtype == SMX_BF16 ? 2 : 4);
  if (rc != SMX_OK) return rc;
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  if (dtype == SMX_BF16) return expdecay_impl<bf16_t>(S, lds, out, ldo, B, T, D, decay, 0, workspace, STREAM);
  return expdecay_impl<float>(S, lds, out, ldo, B, T, D, decay, 0, workspace, STREAM);
}
extern "C" int smx_expdecay_mean_bwd(int dtype, const void* dOut, int64_t ldo, void* dS, int64_t lds, int B, int T, int D,
                                     float decay, void* workspace, void* stream) {
  int rc = expdecay_check(dOut, ldo, dS, lds, D, decay, workspace, dtype == SMX_BF16 ? 2 : 4);
  if (rc != SMX_OK) return rc;
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  if (dtype == SMX_BF16) return expdecay_impl<bf16_t>(dOut, ldo, dS, lds, B, T, D, decay, 1, workspace, STREAM);
  return expdecay_impl<float>(dOut, ldo, dS, lds, B, T, D, decay, 1, workspace, STREAM);
}

// The expdecay summary on a sequence-parallel shard (frames [t_off, t_off + T) of T_glob): the filter crosses the boundary through ONE
// (B, D) state per direction.  phase 1: ends (2, B, D) <- the states LEAVING the shard (f at its last frame, g at its first) for zero
// entering states; the caller gathers them and folds f_in = ends_q[0] + decay^T f_in over the earlier shards (g_in over the later
// ones) - (B, D) arithmetic - into `ends`; phase 2 (same workspace): out = the operator with those ENTERING states and the
// denominators of the GLOBAL frame index (summary_mixing.py:316-365; mode 0 forward, 1 the transposed operator M (s / rowsum(M))).
extern "C" int smx_expdecay_mean_sharded(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D, float decay,
                                         int mode, int t_off, int T_glob, int phase, float* ends, void* workspace, void* stream) {
  SMX_REQUIRE(S && workspace && ends && (phase == 1 || phase == 2) && (mode == 0 || mode == 1) && t_off >= 0 && T_glob >= t_off + T,
              "smx_expdecay_mean_sharded: bad arguments");
  SMX_REQUIRE(decay > 0.f && decay < 1.f, "smx_expdecay_mean_sharded: 0 < decay < 1");
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  SMX_REQUIRE(D % 4 == 0 && lds % 4 == 0 && (reinterpret_cast<uintptr_t>(S) % (4 * es)) == 0 && aligned16(workspace) && aligned16(ends) &&
                  (phase == 1 || (out && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) % (4 * es)) == 0)),
              "smx_expdecay_mean_sharded: needs D, leading dimensions multiples of 4 and aligned pointers");
  if (B <= 0 || T <= 0 || D <= 0) return SMX_OK;
  if (dtype == SMX_BF16) return expdecay_impl<bf16_t>(S, lds, out, ldo, B, T, D, decay, mode, workspace, STREAM, phase, t_off, T_glob, ends);
  return expdecay_impl<float>(S, lds, out, ldo, B, T, D, decay, mode, workspace, STREAM, phase, t_off, T_glob, ends);
}


extern "C" int smx_layernorm_fwd(int dtype, const void* X, int64_t ldx, const float* gamma, const float* beta, void* Y,
                                 int64_t ldy, float* stats, int N, int D, float eps, int act, void* stream) {
  SMX_REQUIRE(X && Y && gamma && beta && N >= 0 && D > 0, "smx_layernorm_fwd: bad arguments");
  if (N == 0) return SMX_OK;
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  auto ok = [&](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) % (4 * es)) == 0 && ld % 4 == 0; };
  const bool vec = D % 4 == 0 && ok(X, ldx) && ok(Y, ldy) && aligned16(gamma) && aligned16(beta);
  if (vec && dtype == SMX_BF16 && D > SMX_LN_WG8_FROM && D <= 2048 && D % 8 == 0 && aligned16(X) && aligned16(Y) && ldx % 8 == 0 && ldy % 8 == 0) {
    // mid-width rows: one workgroup per row, 16-byte accesses (layernorm_fwd_wg8_kernel)
    int blocks = (N + SMX_LN_WG8_UF - 1) / SMX_LN_WG8_UF;
    if (blocks > SMX_LN_WG8_FBLOCKS) blocks = SMX_LN_WG8_FBLOCKS;
    hipLaunchKernelGGL((layernorm_fwd_wg8_kernel<SMX_LN_WG8_UF>), dim3(blocks), dim3(256), 0, STREAM, (const bf16_t*)X, ldx, gamma, beta, (bf16_t*)Y, ldy, stats, N, D, eps, act);
    return check_launch("smx_layernorm_fwd");
  }
  if (vec && D <= 2048) {
    const int ch = (D + 255) / 256;
    const int U = ch <= 1 ? 4 : (ch <= 2 ? 2 : 1);
    int blocks = (N + 4 * U - 1) / (4 * U);
    if (blocks > 2048) blocks = 2048;
#define LN_FWD(TT, CH_, U_) hipLaunchKernelGGL((layernorm_fwd_fast<TT, CH_, U_>), dim3(blocks), dim3(256), 0, STREAM, (const TT*)X, ldx, gamma, beta, (TT*)Y, ldy, stats, N, D, eps, act)
#define LN_FWD_T(TT) do { if (ch <= 1) LN_FWD(TT, 1, 4); else if (ch <= 2) LN_FWD(TT, 2, 2); else if (ch <= 4) LN_FWD(TT, 4, 1); else LN_FWD(TT, 8, 1); } while (0)
    if (dtype == SMX_BF16) LN_FWD_T(bf16_t); else LN_FWD_T(float);
#undef LN_FWD_T
#undef LN_FWD
    return check_launch("smx_layernorm_fwd");
  }
  dim3 grid((N + 3) / 4);
  if (dtype == SMX_BF16) {
    if (vec) hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, true>), grid, dim3(256), 0, STREAM, (const bf16_t*)X, ldx, gamma, beta, (bf16_t*)Y, ldy, stats, N, D, eps, act);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, STREAM, (const bf16_t*)X, ldx, gamma, beta, (bf16_t*)Y, ldy, stats, N, D, eps, act);
  } else {
    if (vec) hipLaunchKernelGGL((layernorm_fwd_kernel<float, true>), grid, dim3(256), 0, STREAM, (const float*)X, ldx, gamma, beta, (float*)Y, ldy, stats, N, D, eps, act);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<float, false>), grid, dim3(256), 0, STREAM, (const float*)X, ldx, gamma, beta, (float*)Y, ldy, stats, N, D, eps, act);
  }
  return check_launch("smx_layernorm_fwd");
}

extern "C" int smx_layernorm_fwd_x32(int dtype, const float* X, int64_t ldx, const float* gamma, const float* beta, void* Y,
                                     int64_t ldy, float* stats, int N, int D, float eps, int act, void* stream) {
  SMX_REQUIRE(X && Y && gamma && beta && N >= 0 && D > 0, "smx_layernorm_fwd_x32: bad arguments");
  if (dtype == SMX_F32) return smx_layernorm_fwd(dtype, X, ldx, gamma, beta, Y, ldy, stats, N, D, eps, act, stream);
  SMX_REQUIRE(dtype == SMX_BF16, "smx_layernorm_fwd_x32: bad dtype");
  if (N == 0) return SMX_OK;
  const bool vec = D % 4 == 0 && D <= 2048 && aligned16(X) && ldx % 4 == 0 && aligned8(Y) && ldy % 4 == 0 && aligned16(gamma) && aligned16(beta);
  if (!vec) return fail(SMX_EUNSUPPORTED, "smx_layernorm_fwd_x32: needs D %% 4 == 0, D <= 2048 and aligned rows");
  const int ch = (D + 255) / 256;
  const int U = ch <= 1 ? 4 : (ch <= 2 ? 2 : 1);
  int blocks = (N + 4 * U - 1) / (4 * U);
  if (blocks > 2048) blocks = 2048;
#define LN_FWDX(CH_, U_) hipLaunchKernelGGL((layernorm_fwd_fast<bf16_t, CH_, U_, float>), dim3(blocks), dim3(256), 0, STREAM, X, ldx, gamma, beta, (bf16_t*)Y, ldy, stats, N, D, eps, act)
  if (ch <= 1) LN_FWDX(1, 4); else if (ch <= 2) LN_FWDX(2, 2); else if (ch <= 4) LN_FWDX(4, 1); else LN_FWDX(8, 1);
#undef LN_FWDX
  return check_launch("smx_layernorm_fwd_x32");
}

extern "C" int smx_layernorm_fwd_pair_x32(int dtype2, const float* X, int64_t ldx, const float* gamma1, const float* beta1, float eps1,
                                          float* Y1, int64_t ldy1, float* stats1, const float* gamma2, const float* beta2,
                                          float eps2, void* Y2, int64_t ldy2, float* stats2, int N, int D, void* stream) {
  SMX_REQUIRE(X && Y1 && Y2 && gamma1 && beta1 && gamma2 && beta2 && N >= 0 && D > 0, "smx_layernorm_fwd_pair_x32: bad arguments");
  SMX_REQUIRE(dtype2 == SMX_BF16 || dtype2 == SMX_F32, "smx_layernorm_fwd_pair_x32: bad dtype");
  if (N == 0) return SMX_OK;
  const bool vec = D % 4 == 0 && D <= 2048 && aligned16(X) && ldx % 4 == 0 && aligned16(Y1) && ldy1 % 4 == 0 &&
                   (dtype2 == SMX_BF16 ? aligned8(Y2) : aligned16(Y2)) && ldy2 % 4 == 0 && aligned16(gamma1) && aligned16(beta1) &&
                   aligned16(gamma2) && aligned16(beta2);
  if (!vec) return fail(SMX_EUNSUPPORTED, "smx_layernorm_fwd_pair_x32: needs D %% 4 == 0, D <= 2048 and aligned rows");
  const int ch = (D + 255) / 256;
  const int U = ch <= 1 ? 4 : (ch <= 2 ? 2 : 1);          // (the launch geometry of layernorm_fwd_fast: bit-identical sums)
  int blocks = (N + 4 * U - 1) / (4 * U);
  if (blocks > 2048) blocks = 2048;
#define LN_PAIR(TT, CH_, U_) hipLaunchKernelGGL((layernorm_fwd_pair_fast<TT, CH_, U_>), dim3(blocks), dim3(256), 0, STREAM, X, ldx, gamma1, beta1, eps1, Y1, ldy1, stats1, gamma2, beta2, eps2, (TT*)Y2, ldy2, stats2, N, D)
#define LN_PAIR_T(TT) do { if (ch <= 1) LN_PAIR(TT, 1, 4); else if (ch <= 2) LN_PAIR(TT, 2, 2); else if (ch <= 4) LN_PAIR(TT, 4, 1); else LN_PAIR(TT, 8, 1); } while (0)
  if (dtype2 == SMX_BF16) LN_PAIR_T(bf16_t); else LN_PAIR_T(float);
#undef LN_PAIR_T
#undef LN_PAIR
  return check_launch("smx_layernorm_fwd_pair_x32");
}

#ifndef SMX_LNB_BLOCKS
#define SMX_LNB_BLOCKS 1024
#endif
static int ln_bwd_blocks(int N) {
  int blocks = (N + 3) / 4;                              // one row per wave and pass when the rows allow it (D <= 512 keeps ONE row in flight)
  return blocks > SMX_LNB_BLOCKS ? SMX_LNB_BLOCKS : (blocks < 1 ? 1 : blocks);
}

#ifndef SMX_LNB_U1
#define SMX_LNB_U1 2      // rows in flight per wave for D <= 256
#endif
#ifndef SMX_LNB_U2
#define SMX_LNB_U2 1      // ... for 256 < D <= 512 (104 registers = 4 waves per SIMD = the whole 1024-block grid resident; two rows in flight: 75 -> 62 us at 64000 x 512, tools/rowkernels_bench.py)
#endif
template <typename T>
static int ln_bwd_impl(const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma, const float* beta,
                       int act, const float* stats,
                       const void* R, int64_t ldr, void* dX, int64_t lddx, float* dgamma, float* dbeta, int N, int D,
                       float* partial, hipStream_t s, LnSecond sec) {
  auto ok = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T))) == 0 && ld % 4 == 0); };
  const bool vec = D % 4 == 0 && ok(dY, lddy) && ok(X, ldx) && ok(R, ldr) && ok(dX, lddx) && ok(sec.dX2, sec.ld);
  if (sec.dX2 && D > 2048) return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd2: the second output needs D <= 2048");
  const int blocks = ln_bwd_blocks(N);
  dim3 grid(blocks);
  if constexpr (sizeof(T) == 2) {
    auto ok16 = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % 16) == 0 && ld % 8 == 0); };
    if (vec && !sec.dX2 && D > SMX_LN_WG8_FROM && D <= 2048 && D % 8 == 0 && ok16(dY, lddy) && ok16(X, ldx) && ok16(R, ldr) && ok16(dX, lddx)) {
      hipLaunchKernelGGL((layernorm_bwd_wg8_kernel<false, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)X, ldx, gamma, beta, act, stats,
                         (const bf16_t*)R, ldr, (bf16_t*)dX, lddx, partial, N, D, (const bf16_t*)nullptr, 0, SMX_ACT_NONE);
      if (dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, partial, blocks, D, dgamma, dbeta);
      return check_launch("smx_layernorm_bwd");
    }
  }
#define LN_BWD(VW, CH) hipLaunchKernelGGL((layernorm_bwd_kernel<T, VW, CH, (VW == 4 && CH == 1 ? SMX_LNB_U1 : (CH <= 2 ? SMX_LNB_U2 : 1))>), grid, dim3(256), 0, s, (const T*)dY, lddy, (const T*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D, sec)
  if (vec) {
    if (D <= 256) LN_BWD(4, 1);
    else if (D <= 512) LN_BWD(4, 2);
    else if (D <= 1024) LN_BWD(4, 4);
    else if (D <= 2048) LN_BWD(4, 8);
    else if (D <= 3072) hipLaunchKernelGGL((layernorm_bwd_wide4_kernel<T, 3>), grid, dim3(256), 0, s, (const T*)dY, lddy, (const T*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D);
    else if (D <= 4096) hipLaunchKernelGGL((layernorm_bwd_wide4_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)dY, lddy, (const T*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D);
    else if (D <= 4096) hipLaunchKernelGGL((layernorm_bwd_wide_kernel<T, 16>), grid, dim3(256), 0, s, (const T*)dY, lddy, (const T*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D);
    else return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd: D=%d > 4096", D);
  } else {
    if (D <= 256) LN_BWD(1, 4);
    else if (D <= 1024) LN_BWD(1, 16);
    else if (D <= 2048) LN_BWD(1, 32);
    else if (D <= 4096) hipLaunchKernelGGL((layernorm_bwd_wide_kernel<T, 16>), grid, dim3(256), 0, s, (const T*)dY, lddy, (const T*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D);
    else return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd: D=%d > 4096", D);
  }
#undef LN_BWD
  if (dgamma)   // (NULL dgamma/dbeta: the partial rows stay in the workspace for a deferred smx_reduce_jobs)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, partial, blocks, D, dgamma, dbeta);
  return check_launch("smx_layernorm_bwd");
}

extern "C" int smx_layernorm_bwd2(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                                  const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX, int64_t lddx, float* dgamma,
                                  float* dbeta, int N, int D, void* workspace, void* dX2, int64_t lddx2, float alpha2,
                                  const uint8_t* row_mask2, float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(dY && X && gamma && beta && stats && dX && workspace && D > 0 && ((dgamma == nullptr) == (dbeta == nullptr)),
              "smx_layernorm_bwd: bad arguments");
  SMX_REQUIRE(drop_p2 >= 0.f && drop_p2 < 1.f, "smx_layernorm_bwd2: 0 <= drop_p < 1");
  if (N == 0) return SMX_OK;
  LnSecond sec;
  sec.slabs = nullptr; sec.nslab = 0; sec.slab_stride = 0;
  sec.dX2 = dX2; sec.ld = lddx2; sec.alpha = alpha2; sec.mask = row_mask2;
  sec.thresh = (uint32_t)((double)drop_p2 * 4294967296.0); sec.scale = 1.f / (1.f - drop_p2); sec.seed = drop_seed2; sec.epoch = epoch;
  if (dtype == SMX_BF16) return ln_bwd_impl<bf16_t>(dY, lddy, X, ldx, gamma, beta, act, stats, R, ldr, dX, lddx, dgamma, dbeta, N, D, (float*)workspace, STREAM, sec);
  return ln_bwd_impl<float>(dY, lddy, X, ldx, gamma, beta, act, stats, R, ldr, dX, lddx, dgamma, dbeta, N, D, (float*)workspace, STREAM, sec);
}
extern "C" int smx_layernorm_bwd2_x32(int dtype, const void* dY, int64_t lddy, const float* X, int64_t ldx, const float* gamma,
                                      const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX, int64_t lddx,
                                      float* dgamma, float* dbeta, int N, int D, void* workspace, void* dX2, int64_t lddx2, float alpha2,
                                      const uint8_t* row_mask2, float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream) {
  if (dtype == SMX_F32)
    return smx_layernorm_bwd2(dtype, dY, lddy, X, ldx, gamma, beta, act, stats, R, ldr, dX, lddx, dgamma, dbeta, N, D, workspace, dX2, lddx2,
                              alpha2, row_mask2, drop_p2, drop_seed2, epoch, stream);
  SMX_REQUIRE(dtype == SMX_BF16 && dY && X && gamma && beta && stats && dX && workspace && D > 0 && ((dgamma == nullptr) == (dbeta == nullptr)),
              "smx_layernorm_bwd2_x32: bad arguments");
  SMX_REQUIRE(drop_p2 >= 0.f && drop_p2 < 1.f, "smx_layernorm_bwd2_x32: 0 <= drop_p < 1");
  if (N == 0) return SMX_OK;
  typedef bf16_t T;
  auto ok = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % 8) == 0 && ld % 4 == 0); };
  const bool vec = D % 4 == 0 && D <= 2048 && ok(dY, lddy) && aligned16(X) && ldx % 4 == 0 && ok(R, ldr) && ok(dX, lddx) && ok(dX2, lddx2);
  if (!vec) return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd2_x32: needs D %% 4 == 0, D <= 2048 and aligned rows");
  LnSecond sec;
  sec.slabs = nullptr; sec.nslab = 0; sec.slab_stride = 0;
  sec.dX2 = dX2; sec.ld = lddx2; sec.alpha = alpha2; sec.mask = row_mask2;
  sec.thresh = (uint32_t)((double)drop_p2 * 4294967296.0); sec.scale = 1.f / (1.f - drop_p2); sec.seed = drop_seed2; sec.epoch = epoch;
  const int blocks = ln_bwd_blocks(N);
  dim3 grid(blocks);
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t s = STREAM;
  {
    auto ok16 = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % 16) == 0 && ld % 8 == 0); };
    if (!dX2 && D > SMX_LN_WG8_FROM && D % 8 == 0 && ok16(dY, lddy) && ldx % 8 == 0 && ok16(R, ldr) && ok16(dX, lddx)) {
      hipLaunchKernelGGL((layernorm_bwd_wg8_kernel<false, float>), grid, dim3(256), 0, s, (const bf16_t*)dY, lddy, X, ldx, gamma, beta, act, stats,
                         (const bf16_t*)R, ldr, (bf16_t*)dX, lddx, partial, N, D, (const bf16_t*)nullptr, 0, SMX_ACT_NONE);
      if (dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, partial, blocks, D, dgamma, dbeta);
      return check_launch("smx_layernorm_bwd2_x32");
    }
  }
#define LN_BWDX(CH) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 4, CH, (CH == 1 ? SMX_LNB_U1 : (CH <= 2 ? SMX_LNB_U2 : 1)), float>), grid, dim3(256), 0, s, (const T*)dY, lddy, X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D, sec)
  if (D <= 256) LN_BWDX(1); else if (D <= 512) LN_BWDX(2); else if (D <= 1024) LN_BWDX(4); else LN_BWDX(8);
#undef LN_BWDX
  if (dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, partial, blocks, D, dgamma, dbeta);
  return check_launch("smx_layernorm_bwd2_x32");
}
extern "C" int smx_layernorm_bwd(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                                 const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX, int64_t lddx, float* dgamma,
                                 float* dbeta, int N, int D, void* workspace, void* stream) {
  return smx_layernorm_bwd2(dtype, dY, lddy, X, ldx, gamma, beta, act, stats, R, ldr, dX, lddx, dgamma, dbeta, N, D, workspace, nullptr, 0,
                            1.f, nullptr, 0.f, 0, nullptr, stream);
}

// The LayerNorm backward whose incoming gradient is the SUM of float32 split-K slabs (smx_gemm_panel_slabs: the dgrad of the Linear
// behind the LayerNorm, K cut into nslab slices): one launch instead of reducer + LayerNorm backward.  x_f32: the LayerNorm input is
// the float32 residual stream (else dtype T).  Everything else as smx_layernorm_bwd2 (res, second output, fused activation).
extern "C" int smx_layernorm_bwd2_slabs(int dtype, const float* slabs, int nslab, int64_t slab_stride, const void* X, int64_t ldx, int x_f32,
                                        const float* gamma, const float* beta, int act, const float* stats, const void* R, int64_t ldr,
                                        void* dX, int64_t lddx, int N, int D, void* workspace, void* dX2, int64_t lddx2, float alpha2,
                                        const uint8_t* row_mask2, float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(dtype == SMX_BF16 && slabs && nslab >= 1 && nslab <= 16 && X && gamma && beta && stats && dX && workspace && D > 0,
              "smx_layernorm_bwd2_slabs: bad arguments");
  SMX_REQUIRE(drop_p2 >= 0.f && drop_p2 < 1.f, "smx_layernorm_bwd2_slabs: 0 <= drop_p < 1");
  if (N == 0) return SMX_OK;
  typedef bf16_t T;
  auto ok = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % 8) == 0 && ld % 4 == 0); };
  const bool vec = D % 4 == 0 && D <= 2048 && aligned16(slabs) && slab_stride % 4 == 0 && (x_f32 ? aligned16(X) : ok(X, ldx)) && ldx % 4 == 0 &&
                   ok(R, ldr) && ok(dX, lddx) && ok(dX2, lddx2);
  if (!vec) return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd2_slabs: needs D %% 4 == 0, D <= 2048 and aligned rows");
  LnSecond sec;
  sec.slabs = slabs; sec.nslab = nslab; sec.slab_stride = slab_stride;
  sec.dX2 = dX2; sec.ld = lddx2; sec.alpha = alpha2; sec.mask = row_mask2;
  sec.thresh = (uint32_t)((double)drop_p2 * 4294967296.0); sec.scale = 1.f / (1.f - drop_p2); sec.seed = drop_seed2; sec.epoch = epoch;
  const int blocks = ln_bwd_blocks(N);
  dim3 grid(blocks);
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t s = STREAM;
#define LN_BWDS(CH, TXX) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 4, CH, 1, TXX>), grid, dim3(256), 0, s, (const T*)nullptr, 0, (const TXX*)X, ldx, gamma, beta, act, stats, (const T*)R, ldr, (T*)dX, lddx, partial, N, D, sec)
  if (x_f32) { if (D <= 256) LN_BWDS(1, float); else if (D <= 512) LN_BWDS(2, float); else if (D <= 1024) LN_BWDS(4, float); else LN_BWDS(8, float); }
  else { if (D <= 256) LN_BWDS(1, T); else if (D <= 512) LN_BWDS(2, T); else if (D <= 1024) LN_BWDS(4, T); else LN_BWDS(8, T); }
#undef LN_BWDS
  return check_launch("smx_layernorm_bwd2_slabs");
}

// dZ = zact'(Z) * (LayerNorm backward of dY), for a LayerNorm whose input is X = zact(Z) (the CSGU norm of the cgMLP: X = the gate
// half of GELU(channel_proj1(.)), Branchformer.py:84-96 of the reference's ConvolutionBranch).  bf16, D <= 2048, D % 8 == 0,
// 16-byte aligned rows; dgamma / dbeta NULL = the partial rows stay in the workspace (smx_layernorm_bwd_workspace bytes).
extern "C" int smx_layernorm_bwd_preact(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                                        const float* beta, int act, const float* stats, const void* Z, int64_t ldz, int zact,
                                        void* dX, int64_t lddx, float* dgamma, float* dbeta, int N, int D, void* workspace, void* stream) {
  SMX_REQUIRE(dY && X && gamma && beta && stats && Z && dX && workspace && D > 0 && ((dgamma == nullptr) == (dbeta == nullptr)),
              "smx_layernorm_bwd_preact: bad arguments");
  if (N == 0) return SMX_OK;
  auto ok16 = [&](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0 && ld % 8 == 0; };
  if (act != SMX_ACT_NONE) return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd_preact: a LayerNorm without a fused activation of its own");
  if (dtype != SMX_BF16 || D > 2048 || D % 8 != 0 || !ok16(dY, lddy) || !ok16(X, ldx) || !ok16(Z, ldz) || !ok16(dX, lddx))
    return fail(SMX_EUNSUPPORTED, "smx_layernorm_bwd_preact: needs bf16, D <= 2048, D %% 8 == 0 and 16-byte aligned rows");
  const int blocks = ln_bwd_blocks(N);
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t s = STREAM;
  hipLaunchKernelGGL((layernorm_bwd_wg8_kernel<true, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)X, ldx, gamma, beta,
                     act, stats, (const bf16_t*)nullptr, 0, (bf16_t*)dX, lddx, partial, N, D, (const bf16_t*)Z, ldz, zact);
  if (dgamma) hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, partial, blocks, D, dgamma, dbeta);
  return check_launch("smx_layernorm_bwd_preact");
}

extern "C" int smx_layernorm_bwd_blocks(int N) { return ln_bwd_blocks(N); }
extern "C" size_t smx_layernorm_bwd_workspace(int N, int D) { return (size_t)ln_bwd_blocks(N) * 2 * D * sizeof(float); }

static const int ACT_BWD_RS = 32;
static const int ACT_BWD_YMAX = 512;   // row-strip workgroups per column block (= partial rows of the bias reduction)
extern "C" size_t smx_act_mask_bwd_workspace(int N, int M) { return (size_t)((N + ACT_BWD_RS - 1) / ACT_BWD_RS) * M * sizeof(float); }

extern "C" int smx_act_mask_bwd(int dtype, const void* dY, int64_t lddy, const void* Z, int64_t ldz,
                                const uint8_t* row_mask, void* dZ, int64_t lddz, int N, int M, int act, float alpha,
                                float* dbias, float* dgroup, int64_t lddgroup, int group_div, float drop_p,
                                uint64_t drop_seed, const uint64_t* epoch, void* workspace, void* stream) {
  SMX_REQUIRE(dY && N >= 0 && M > 0, "smx_act_mask_bwd: bad arguments");
  SMX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "smx_act_mask_bwd: 0 <= drop_p < 1");
  const uint32_t dthresh = (uint32_t)((double)drop_p * 4294967296.0);
  const float dscale = 1.f / (1.f - drop_p);
  SMX_REQUIRE(!dgroup || group_div > 0, "smx_act_mask_bwd: group_div must be > 0");
  if (N == 0) return SMX_OK;
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  const int nv = dtype == SMX_BF16 ? 8 : 4;
  auto ok = [&](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && (ld * (int64_t)es) % 16 == 0); };
  const bool vec = M % nv == 0 && ok(dY, lddy) && ok(Z, ldz) && ok(dZ, lddz) && (!dbias || workspace);
  if (vec) {
    int chunks = M / nv, LPR = 1;
    while (LPR < 64 && LPR < chunks) LPR <<= 1;       // lanes per row chunk (power of two <= 64)
    const int RS = ACT_BWD_RS;
    float* partial = reinterpret_cast<float*>(workspace);
    const int ny = min((N + RS - 1) / RS, ACT_BWD_YMAX);
    dim3 grid((chunks + LPR - 1) / LPR, ny);
    if (dtype == SMX_BF16) hipLaunchKernelGGL((act_mask_bwd_vec_kernel<bf16_t>), grid, dim3(256), 0, STREAM, (const bf16_t*)dY, lddy, (const bf16_t*)Z, ldz, row_mask, (bf16_t*)dZ, lddz, N, M, act, alpha, dbias, dgroup, lddgroup, group_div, RS, LPR, partial, dthresh, dscale, drop_seed, epoch);
    else hipLaunchKernelGGL((act_mask_bwd_vec_kernel<float>), grid, dim3(256), 0, STREAM, (const float*)dY, lddy, (const float*)Z, ldz, row_mask, (float*)dZ, lddz, N, M, act, alpha, dbias, dgroup, lddgroup, group_div, RS, LPR, partial, dthresh, dscale, drop_seed, epoch);
    if (dbias) hipLaunchKernelGGL(colsum_partials_kernel, dim3((M + 15) / 16), dim3(256), 0, STREAM, partial, ny, M, dbias);
  } else {
    const int RS = 128;
    dim3 grid((M + 255) / 256, (N + RS - 1) / RS);
    if (dtype == SMX_BF16) hipLaunchKernelGGL((act_mask_bwd_kernel<bf16_t>), grid, dim3(256), 0, STREAM, (const bf16_t*)dY, lddy, (const bf16_t*)Z, ldz, row_mask, (bf16_t*)dZ, lddz, N, M, act, alpha, dbias, dgroup, lddgroup, group_div, RS, dthresh, dscale, drop_seed, epoch);
    else hipLaunchKernelGGL((act_mask_bwd_kernel<float>), grid, dim3(256), 0, STREAM, (const float*)dY, lddy, (const float*)Z, ldz, row_mask, (float*)dZ, lddz, N, M, act, alpha, dbias, dgroup, lddgroup, group_div, RS, dthresh, dscale, drop_seed, epoch);
  }
  return check_launch("smx_act_mask_bwd");
}

extern "C" int smx_axpby(int dtype, float a, const void* X, int64_t ldx, float b, const void* Y0, int64_t ldy0, void* Y,
                         int64_t ldy, int N, int D, void* stream) {
  SMX_REQUIRE(X && Y, "smx_axpby: null pointer");
  if (N <= 0 || D <= 0) return SMX_OK;
  const size_t es = dtype == SMX_BF16 ? 2 : 4;
  auto ok = [&](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) % (4 * es)) == 0 && ld % 4 == 0); };
  const bool vec = ok(X, ldx) && ok(Y0, ldy0) && ok(Y, ldy);
  int grid = grid1d((long)N * ((D + 3) / 4));
  if (dtype == SMX_BF16) {
    if (vec) hipLaunchKernelGGL((axpby_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, STREAM, a, (const bf16_t*)X, ldx, b, (const bf16_t*)Y0, ldy0, (bf16_t*)Y, ldy, N, D);
    else hipLaunchKernelGGL((axpby_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, STREAM, a, (const bf16_t*)X, ldx, b, (const bf16_t*)Y0, ldy0, (bf16_t*)Y, ldy, N, D);
  } else {
    if (vec) hipLaunchKernelGGL((axpby_kernel<float, true>), dim3(grid), dim3(256), 0, STREAM, a, (const float*)X, ldx, b, (const float*)Y0, ldy0, (float*)Y, ldy, N, D);
    else hipLaunchKernelGGL((axpby_kernel<float, false>), dim3(grid), dim3(256), 0, STREAM, a, (const float*)X, ldx, b, (const float*)Y0, ldy0, (float*)Y, ldy, N, D);
  }
  return check_launch("smx_axpby");
}

extern "C" int smx_dropout(int dtype, const void* X, int64_t ldx, void* Y, int64_t ldy, int N, int D, float p,
                           uint64_t seed, const uint64_t* epoch, void* stream) {
  SMX_REQUIRE(X && Y && p >= 0.f && p < 1.f, "smx_dropout: bad arguments (0 <= p < 1)");
  if (N <= 0 || D <= 0) return SMX_OK;
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  int grid = grid1d((long)N * ((D + 3) / 4));
  if (dtype == SMX_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(grid), dim3(256), 0, STREAM, (const bf16_t*)X, ldx, (bf16_t*)Y, ldy, N, D, thresh, scale, seed, epoch);
  else hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid), dim3(256), 0, STREAM, (const float*)X, ldx, (float*)Y, ldy, N, D, thresh, scale, seed, epoch);
  return check_launch("smx_dropout");
}

extern "C" int smx_add_rowtable(int dtype, void* Y, int64_t ldy, const float* table, int R, int N, int D, void* stream) {
  SMX_REQUIRE(Y && table && R > 0, "smx_add_rowtable: bad arguments");
  if (N <= 0 || D <= 0) return SMX_OK;
  int grid = grid1d((long)N * D);
  if (dtype == SMX_BF16) hipLaunchKernelGGL((add_rowtable_kernel<bf16_t>), dim3(grid), dim3(256), 0, STREAM, (bf16_t*)Y, ldy, table, R, N, D);
  else hipLaunchKernelGGL((add_rowtable_kernel<float>), dim3(grid), dim3(256), 0, STREAM, (float*)Y, ldy, table, R, N, D);
  return check_launch("smx_add_rowtable");
}

extern "C" int smx_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream) {
  SMX_REQUIRE(src && dst, "smx_cast_from_f32: null pointer");
  if (n <= 0) return SMX_OK;
  if (dtype == SMX_BF16) hipLaunchKernelGGL((cast_from_f32_kernel<bf16_t>), dim3(grid1d(n)), dim3(256), 0, STREAM, src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(grid1d(n)), dim3(256), 0, STREAM, src, (float*)dst, n);
  return check_launch("smx_cast_from_f32");
}
extern "C" int smx_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream) {
  SMX_REQUIRE(src && dst, "smx_cast_to_f32: null pointer");
  if (n <= 0) return SMX_OK;
  if (dtype == SMX_BF16) hipLaunchKernelGGL((cast_to_f32_kernel<bf16_t>), dim3(grid1d(n)), dim3(256), 0, STREAM, (const bf16_t*)src, dst, n);
  else hipLaunchKernelGGL((cast_to_f32_kernel<float>), dim3(grid1d(n)), dim3(256), 0, STREAM, (const float*)src, dst, n);
  return check_launch("smx_cast_to_f32");
}

extern "C" int smx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              float grad_scale, const float* gscale_dev, const uint64_t* step_dev, void* stream) {
  SMX_REQUIRE(param && grad && exp_avg && exp_avg_sq, "smx_adamw_step: bad arguments");
  if (n <= 0) return SMX_OK;
  SMX_REQUIRE(step > 0 || step_dev, "smx_adamw_step: step <= 0 needs the device step counter (step_dev)");
  float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid1d(n)), dim3(256), 0, STREAM, param, grad, exp_avg, exp_avg_sq,
                     (uint16_t*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, gscale_dev,
                     step > 0 ? nullptr : step_dev);
  return check_launch("smx_adamw_step");
}
extern "C" size_t smx_sumsq_workspace(void) { return SUMSQ_BLOCKS * sizeof(float); }
extern "C" int smx_sumsq(const float* x, int64_t n, float* out, void* workspace, void* stream) {
  SMX_REQUIRE(x && out && workspace, "smx_sumsq: null pointer");
  if (n <= 0) return SMX_OK;
  int g = grid1d(n);
  if (g > SUMSQ_BLOCKS) g = SUMSQ_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(g), dim3(256), 0, STREAM, x, n, reinterpret_cast<float*>(workspace));
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, STREAM, reinterpret_cast<const float*>(workspace), g, out);
  return check_launch("smx_sumsq");
}
extern "C" int smx_clip_factor(const float* sumsq, float max_norm, float inv_scale, float* out, void* stream) {
  SMX_REQUIRE(sumsq && out, "smx_clip_factor: null pointer");
  hipLaunchKernelGGL(clip_factor_kernel, dim3(1), dim3(1), 0, STREAM, sumsq, max_norm, inv_scale, out);
  return check_launch("smx_clip_factor");
}
