// frontend.hip — the step in front of the encoder, gfx950:
//   log-mel filterbank  : frame+window -> [DFT as an exact-fp32 MFMA GEMM, smx_gemm] -> power + mel + dB -> top_db clamp
//   conv subsampling    : im2col (3x3, stride 2, reflect pad 1, channels-last) -> [smx_gemm + bias] -> LayerNorm over
//                         (F, C) + LeakyReLU (smx_layernorm_fwd with fused activation); col2im for the backward.
// The arithmetic of these stages lives in un-vendored SpeechBrain (Fbank / ConvolutionFrontEnd, recipes/LibriSpeech/
// ASR/transducer/hparams/conformer_summarymixing_transducer.yaml:167-175,247-254): the reference pins nothing here, the
// spec is oracle/smx_oracle.py::fbank / conv_frontend ("parity unpinned").
#include "smx_common.h"

namespace smx {

// frames[b*T + t, j] = window[j] * wav[b, t*hop + j - n_fft/2]   (center=True, zero padding), fp32
__global__ __launch_bounds__(256) void frame_window_kernel(const float* __restrict__ wav, long ldw, const float* __restrict__ win,
                                                           float* __restrict__ out, int B, int L, int T, int n_fft, int hop) {
  const long total = (long)B * T * (n_fft / 4);
  const int half = n_fft / 2, q4 = n_fft / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % q4) * 4;
    const long row = i / q4;
    const int t = (int)(row % T), b = (int)(row / T);
    const long s0 = (long)t * hop + j - half;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long s = s0 + q;
      v[q] = (s >= 0 && s < L) ? wav[(long)b * ldw + s] * win[j + q] : 0.f;
    }
    *reinterpret_cast<float4*>(out + row * n_fft + j) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// One wave per frame, workgroups stride over the frames: P[f] = re^2 + im^2 (LDS), mel[m] = sum_f P[f] * fb[m][f],
// db = 10 log10(max(mel, amin)).  Triangular filters are nonzero on one short bin range each (about 2 * n_bins nonzeros
// in the whole (n_mels, n_bins) matrix): a lane finds the range of its filters once and then only multiplies inside it
// - 40x fewer MACs than the dense product (3.0 ms -> 0.2 ms for 128 000 frames).
// spec layout: S (N, lds) with re at column f and im at column im_off + f.
__global__ __launch_bounds__(256) void mel_db_kernel(const float* __restrict__ S, long lds, int im_off, const float* __restrict__ fb,
                                                     int n_bins, int n_mels, float amin, float* __restrict__ db,
                                                     float* __restrict__ bmax, int N_) {
  extern __shared__ float pw[];                        // 4 x n_bins power rows | n_mels x MELW filter bands
  constexpr int MELW = 48;                             // widest band kept in LDS (n_fft = 512, 80 mels: <= 27 bins); longer tails read global
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* P = pw + w * n_bins;
  float* band = pw + 4 * n_bins;
  constexpr int MPL = 4;                               // filters per lane (n_mels <= 256)
  // band limits of every filter, found by the whole workgroup with coalesced row reads (a lane walking its own filter row was
  // 2 x 257 uncoalesced loads per lane and workgroup: most of this kernel's 0.5 ms)
  __shared__ int slo[256], shi[256];
  for (int m = threadIdx.x; m < 256; m += 256) { slo[m] = n_bins; shi[m] = 0; }
  __syncthreads();
  // (one wave per filter row, two rows at a time: every load of the pair is in flight before the first compare, limits by wave
  //  shuffles - the loop over all rows with an LDS atomic behind every load was 160 dependent round trips per workgroup, a
  //  constant ~60 us in front of the first frame: 94 us of kernel for the recipe batch's 15 000 frames)
  for (int m0 = 2 * w; m0 < n_mels; m0 += 8) {
    int l2[2] = {n_bins, n_bins}, h2[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u;
      if (m < n_mels) {
        const float* fr = fb + (long)m * n_bins;
        for (int f = lane; f < n_bins; f += 64)
          if (fr[f] != 0.f) { l2[u] = min(l2[u], f); h2[u] = max(h2[u], f + 1); }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) { l2[u] = min(l2[u], __shfl_xor(l2[u], off, 64)); h2[u] = max(h2[u], __shfl_xor(h2[u], off, 64)); }
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (m0 + u < n_mels) { slo[m0 + u] = l2[u]; shi[m0 + u] = h2[u]; }
    }
  }
  __syncthreads();
  int lo[MPL], hi[MPL];
#pragma unroll
  for (int k = 0; k < MPL; ++k) {
    const int m = lane + 64 * k;
    lo[k] = n_bins; hi[k] = 0;
    if (m < n_mels) {
      lo[k] = slo[m]; hi[k] = shi[m];
      const float* fr = fb + (long)m * n_bins;
      // the band's weights are the same for every frame: once into LDS (a global load per tap and frame made this kernel
      // latency-bound: 845 us for 256 000 frames)
      for (int j = w; j < MELW; j += 4) band[m * MELW + j] = (lo[k] + j < hi[k]) ? fr[lo[k] + j] : 0.f;   // (each wave a quarter of the taps)
    }
  }
  __syncthreads();
  for (int n = blockIdx.x * 4 + w; n < N_; n += gridDim.x * 4) {
    const float* s = S + (long)n * lds;
    for (int f = lane; f < n_bins; f += 64) { const float re = s[f], im = s[im_off + f]; P[f] = re * re + im * im; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes are visible to all its lanes
    float fmx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < MPL; ++k) {
      const int m = lane + 64 * k;
      if (m < n_mels) {
        const float* fr = fb + (long)m * n_bins;
        const float* bw = band + m * MELW;
        const int wdt = hi[k] - lo[k], wl = min(wdt, MELW);
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        for (; j + 1 < wl; j += 2) { a0 += P[lo[k] + j] * bw[j]; a1 += P[lo[k] + j + 1] * bw[j + 1]; }
        if (j < wl) a0 += P[lo[k] + j] * bw[j];
        for (int f = lo[k] + MELW; f < hi[k]; ++f) a1 += P[f] * fr[f];
        const float dbv = 10.f * log10f(fmaxf(a0 + a1, amin));
        db[(long)n * n_mels + m] = dbv;
        fmx = fmaxf(fmx, dbv);
      }
    }
    // the frame's maximum (the per-utterance top_db clamp needs max over the utterance: utt_max_kernel then reads one value per
    // frame instead of n_mels)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fmx = fmaxf(fmx, __shfl_xor(fmx, off, 64));
    if (lane == 0) bmax[n] = fmx;
    __builtin_amdgcn_wave_barrier();                     // (P is rewritten by the next frame)
  }
}

// umax[b] = max over the blocks of utterance b (blocks never straddle utterances when T % 4 == 0; otherwise the
// per-frame fallback below is used).  One block per utterance.
__global__ __launch_bounds__(256) void utt_max_kernel(const float* __restrict__ db, int T, int n_mels, float* __restrict__ umax) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float mx = -3.0e38f;
  const long base = (long)b * T * n_mels, cnt = (long)T * n_mels;
  for (long i = threadIdx.x; i < cnt; i += 256) mx = fmaxf(mx, db[base + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) umax[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void topdb_clamp_kernel(const float* __restrict__ db, const float* __restrict__ umax, float top_db,
                                                          T* __restrict__ out, long per_utt, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / per_utt);
    out[i] = from_f32<T>(fmaxf(db[i], umax[b] - top_db));
  }
}

__device__ __forceinline__ int reflect1(int i, int L) { return i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i); }

// im2col for a 3x3 / stride 2 / reflect-pad-1 convolution over (time, freq), channels-last.
//   x (B, T, F, C) -> col (B*T2*F2, Kp), column = (dt*3 + df)*C + c, columns >= 9*C are zero.  T2 = ceil(T/2).
template <typename T>
__global__ __launch_bounds__(256) void im2col_s2_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int T_, int F, int C,
                                                        int T2, int F2, int Kp) {
  const long total = (long)B * T2 * F2 * Kp;
  const int K = 9 * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const long row = i / Kp;
    T v = from_f32<T>(0.f);
    if (k < K) {
      const int c = k % C, tap = k / C, dt = tap / 3, df = tap % 3;
      const int f2 = (int)(row % F2);
      const long r2 = row / F2;
      const int t2 = (int)(r2 % T2), b = (int)(r2 / T2);
      const int t = reflect1(2 * t2 + dt - 1, T_), f = reflect1(2 * f2 + df - 1, F);
      v = x[(((long)b * T_ + t) * F + f) * C + c];
    }
    col[i] = v;
  }
}

// col2im: dx[b,t,f,c] = sum over (t2,dt,f2,df) whose reflected source is (t,f) of dcol[(b,t2,f2), (dt*3+df)*C + c]
template <typename T>
__global__ __launch_bounds__(256) void col2im_s2_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int T_, int F, int C,
                                                        int T2, int F2, int Kp) {
  const long total = (long)B * T_ * F * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long r = i / C;
    const int f = (int)(r % F);
    r /= F;
    const int t = (int)(r % T_), b = (int)(r / T_);
    float acc = 0.f;
    // candidate output rows: those whose 3-tap window [2*t2-1, 2*t2+1] touches t directly, plus the reflected edges
    for (int t2 = max(0, (t - 1) / 2 - 1); t2 <= min(T2 - 1, (t + 1) / 2 + 1); ++t2) {
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        if (reflect1(2 * t2 + dt - 1, T_) != t) continue;
        for (int f2 = max(0, (f - 1) / 2 - 1); f2 <= min(F2 - 1, (f + 1) / 2 + 1); ++f2) {
#pragma unroll
          for (int df = 0; df < 3; ++df) {
            if (reflect1(2 * f2 + df - 1, F) != f) continue;
            acc += to_f32(dcol[(((long)b * T2 + t2) * F2 + f2) * Kp + (dt * 3 + df) * C + c]);
          }
        }
      }
    }
    dx[i] = from_f32<T>(acc);
  }
}

// im2col for C % 8 == 0 in 16-byte pieces: one thread = (output row, tap, 8 channels): one 16-byte load, one 16-byte store
// (the element-wise kernel above spends 2.4 ms on the 1.47 GB col of the second block: two integer divisions per element)
__global__ __launch_bounds__(256) void im2col_s2_vec_kernel(const uint4* __restrict__ x, uint4* __restrict__ col, int B, int T_,
                                                            int F, int C8, int T2, int F2, int Kp8) {
  const long rows = (long)B * T2 * F2;
  const int per_row = 9 * C8;                                 // (Kp == 9 C for C % 8 == 0: no padding columns)
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int f2 = (int)(row % F2);
    const long r2 = row / F2;
    const int t2 = (int)(r2 % T2), b = (int)(r2 / T2);
    for (int e = threadIdx.x & 63; e < per_row; e += 64) {
      const int tap = e / C8, c8 = e - tap * C8, dt = tap / 3, df = tap - dt * 3;
      const int t = reflect1(2 * t2 + dt - 1, T_), f = reflect1(2 * f2 + df - 1, F);
      col[row * Kp8 + e] = x[(((long)b * T_ + t) * F + f) * C8 + c8];
    }
  }
}

// ---- first conv block in ONE pass per direction (the recipe's shape: 1 input channel, O = 64 output channels, F2 = F / 2 a
// multiple of 8; 3 x 3, stride 2, reflect pad 1; LayerNorm over the (F2, O) row; activation) ---------------------------------
//   forward : a[b,t2,f2,c] = act(LN_row(conv(x)[b,t2,:,:] + bias) * gamma + beta),  stats[b,t2] = (mean, rstd)
//   backward: from dA and x alone - the convolution is recomputed (9 MACs per output), so neither the pre-LayerNorm
//             tensor nor the patch matrix is ever stored, and since the block's input is the feature map (no gradient
//             wanted) NOTHING of activation size is written: only per-workgroup partial rows
//             [dgamma (F2 O) | dbeta (F2 O) | dW (O x 9) | dbias (O)], reduced in a fixed order by rows_sum_add_kernel.
// One workgroup walks rows (b, t2); thread t owns the channel pair c0 = 2 (t % 32) at f2 = t / 32 + 8 i: its 9-tap weights
// sit in registers, the row's 3 x (F + 2) input window in LDS, a row's 2 NI values per thread in registers between the
// statistics pass and the normalisation.  Replaces im2col (0.26 ms) + the K = 16 Linear (0.30) + LayerNorm (0.42) by one
// 655 MB write, and the wide LayerNorm backward (0.50) + conv-1 wgrad GEMM (0.17) by one 655 MB read (B = 128 x 20 s).
template <typename T, int NI>
__global__ __launch_bounds__(256) void conv1_ln_fwd_kernel(const T* __restrict__ X, const float* __restrict__ W9,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int act,
                                                           T* __restrict__ Y, float* __restrict__ stats, int B, int T_, int F,
                                                           int T2) {
  constexpr int O = 64;
  __shared__ float xs[3][164];                            // (F + 2) <= 162 columns: column fi + 1 = reflected feature fi
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c0 = 2 * (t & 31), fb = t >> 5, F2 = F / 2, D = F2 * O;
  float w0[9], w1[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { w0[j] = W9[c0 * 9 + j]; w1[j] = W9[(c0 + 1) * 9 + j]; }
  const float b0 = bias ? bias[c0] : 0.f, b1 = bias ? bias[c0 + 1] : 0.f;
  float gm[NI][2], bt[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int o = (fb + 8 * i) * O + c0;
    gm[i][0] = gamma[o]; gm[i][1] = gamma[o + 1]; bt[i][0] = beta[o]; bt[i][1] = beta[o + 1];
  }
  const long rows = (long)B * T2;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int b = (int)(row / T2), t2 = (int)(row % T2);
    __syncthreads();
    for (int e = t; e < 3 * (F + 2); e += 256) {
      const int dt = e / (F + 2), fi = e % (F + 2) - 1;
      const int ti = reflect1(2 * t2 + dt - 1, T_), ff = reflect1(fi, F);
      xs[dt][fi + 1] = to_f32(X[((long)b * T_ + ti) * F + ff]);
    }
    __syncthreads();
    float y[NI][2], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int f2 = fb + 8 * i;
      float a0 = b0, a1 = b1;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int df = 0; df < 3; ++df) {
          const float xv = xs[dt][2 * f2 + df];
          a0 += w0[dt * 3 + df] * xv; a1 += w1[dt * 3 + df] * xv;
        }
      y[i][0] = a0; y[i][1] = a1;
      s1 += a0 + a1;
    }
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wv] = s1;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
#pragma unroll
    for (int i = 0; i < NI; ++i) { const float e0 = y[i][0] - mean, e1 = y[i][1] - mean; s2 += e0 * e0 + e1 * e1; }
    s2 = wave_sum(s2);                                     // (two passes over the registers: no E[y^2] - mean^2 cancellation)
    if (lane == 0) red[1][wv] = s2;
    __syncthreads();
    const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
    const float rstd = rsqrtf(var + eps);
    if (t == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    T* yr = Y + row * D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int o = (fb + 8 * i) * O + c0;
      const float v0 = act_fwd(act, (y[i][0] - mean) * rstd * gm[i][0] + bt[i][0]);
      const float v1 = act_fwd(act, (y[i][1] - mean) * rstd * gm[i][1] + bt[i][1]);
      if constexpr (sizeof(T) == 2) *reinterpret_cast<uint32_t*>(yr + o) = pack_bf16x2(v0, v1);
      else *reinterpret_cast<float2*>(yr + o) = make_float2(v0, v1);
    }
  }
}

template <typename T, int NI>
__global__ __launch_bounds__(256) void conv1_ln_bwd_kernel(const T* __restrict__ dA, const T* __restrict__ X,
                                                           const float* __restrict__ W9, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, int act, float* __restrict__ partial,
                                                           int B, int T_, int F, int T2) {
  constexpr int O = 64;
  __shared__ float xs[2][3][164];                         // double-buffered: the next row's window lands while this one is used
  __shared__ float red[2][2][4];
  __shared__ float fold[8][32][20];                       // [f group][channel pair][dW 2 x 9 | db 2]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c0 = 2 * (t & 31), fb = t >> 5, F2 = F / 2, D = F2 * O;
  float w0[9], w1[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { w0[j] = W9[c0 * 9 + j]; w1[j] = W9[(c0 + 1) * 9 + j]; }
  const float b0 = bias ? bias[c0] : 0.f, b1 = bias ? bias[c0 + 1] : 0.f;
  float gm[NI][2], bt[NI][2], dgm[NI][2], dbt[NI][2], dw0[9], dw1[9], db0 = 0.f, db1 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int o = (fb + 8 * i) * O + c0;
    gm[i][0] = gamma[o]; gm[i][1] = gamma[o + 1]; bt[i][0] = beta[o]; bt[i][1] = beta[o + 1];
    dgm[i][0] = dgm[i][1] = dbt[i][0] = dbt[i][1] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) dw0[j] = dw1[j] = 0.f;
  const long rows = (long)B * T2;
  // the loads of row r + stride (its dA values, its window, its statistics) are issued before the arithmetic of row r
  auto window = [&](long row, int buf) {
    const int b = (int)(row / T2), t2 = (int)(row % T2);
    for (int e = t; e < 3 * (F + 2); e += 256) {
      const int dt = e / (F + 2), fi = e % (F + 2) - 1;
      const int ti = reflect1(2 * t2 + dt - 1, T_), ff = reflect1(fi, F);
      xs[buf][dt][fi + 1] = to_f32(X[((long)b * T_ + ti) * F + ff]);
    }
  };
  auto load_da = [&](long row, float (&d)[NI][2]) {
    const T* dr = dA + row * D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int o = (fb + 8 * i) * O + c0;
      if constexpr (sizeof(T) == 2) {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(dr + o);
        d[i][0] = bf16_bits_to_f32(u & 0xffffu); d[i][1] = bf16_bits_to_f32(u >> 16);
      } else {
        const float2 u = *reinterpret_cast<const float2*>(dr + o);
        d[i][0] = u.x; d[i][1] = u.y;
      }
    }
  };
  float dn[NI][2];
  float mean_n = 0.f, rstd_n = 0.f;
  long row = blockIdx.x;
  int buf = 0;
  if (row < rows) { window(row, 0); load_da(row, dn); mean_n = stats[2 * row]; rstd_n = stats[2 * row + 1]; }
  for (; row < rows; row += gridDim.x, buf ^= 1) {
    __syncthreads();                                       // this row's window is complete; the other buffer is free
    float d[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) { d[i][0] = dn[i][0]; d[i][1] = dn[i][1]; }
    const float mean = mean_n, rstd = rstd_n;
    const long nxt = row + gridDim.x;
    if (nxt < rows) { window(nxt, buf ^ 1); load_da(nxt, dn); mean_n = stats[2 * nxt]; rstd_n = stats[2 * nxt + 1]; }
    float xh[NI][2], g[NI][2], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int f2 = fb + 8 * i;
      float a0 = b0, a1 = b1;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int df = 0; df < 3; ++df) {
          const float xv = xs[buf][dt][2 * f2 + df];
          a0 += w0[dt * 3 + df] * xv; a1 += w1[dt * 3 + df] * xv;
        }
      float d0 = d[i][0], d1 = d[i][1];
      xh[i][0] = (a0 - mean) * rstd; xh[i][1] = (a1 - mean) * rstd;
      if (act != SMX_ACT_NONE) {                           // (uniform)
        d0 *= act_grad(act, xh[i][0] * gm[i][0] + bt[i][0]);
        d1 *= act_grad(act, xh[i][1] * gm[i][1] + bt[i][1]);
      }
      dgm[i][0] += d0 * xh[i][0]; dgm[i][1] += d1 * xh[i][1];
      dbt[i][0] += d0; dbt[i][1] += d1;
      g[i][0] = d0 * gm[i][0]; g[i][1] = d1 * gm[i][1];
      s1 += g[i][0] + g[i][1]; s2 += g[i][0] * xh[i][0] + g[i][1] * xh[i][1];
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { red[buf][0][wv] = s1; red[buf][1][wv] = s2; }   // (double-buffered like the window: one barrier per row)
    __syncthreads();
    const float m1 = ((red[buf][0][0] + red[buf][0][1]) + (red[buf][0][2] + red[buf][0][3])) / (float)D;
    const float m2 = ((red[buf][1][0] + red[buf][1][1]) + (red[buf][1][2] + red[buf][1][3])) / (float)D;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int f2 = fb + 8 * i;
      const float dy0 = rstd * (g[i][0] - m1 - xh[i][0] * m2), dy1 = rstd * (g[i][1] - m1 - xh[i][1] * m2);
      db0 += dy0; db1 += dy1;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int df = 0; df < 3; ++df) {
          const float xv = xs[buf][dt][2 * f2 + df];
          dw0[dt * 3 + df] += dy0 * xv; dw1[dt * 3 + df] += dy1 * xv;
        }
    }
  }
  // partial row of this workgroup: [dgamma D | dbeta D | dW O x 9 | dbias O]
  float* pr = partial + (long)blockIdx.x * (2 * D + O * 10);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int o = (fb + 8 * i) * O + c0;
    pr[o] = dgm[i][0]; pr[o + 1] = dgm[i][1]; pr[D + o] = dbt[i][0]; pr[D + o + 1] = dbt[i][1];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 9; ++j) { fold[fb][t & 31][j] = dw0[j]; fold[fb][t & 31][9 + j] = dw1[j]; }
  fold[fb][t & 31][18] = db0; fold[fb][t & 31][19] = db1;
  __syncthreads();
  for (int e = t; e < 32 * 20; e += 256) {                 // the 8 f groups in a fixed order
    const int cp = e / 20, j = e % 20;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += fold[q][cp][j];
    const int c = 2 * cp + (j >= 18 ? j - 18 : j / 9);
    if (j < 18) pr[2 * D + c * 9 + j % 9] = sum; else pr[2 * D + O * 9 + c] = sum;
  }
}

// out[i] += sum_r partial[r][i]  (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void rows_sum_add_kernel(const float* __restrict__ partial, int nrows, long W, float* __restrict__ out) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= W) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < nrows; r += 4) {
    s0 += partial[(long)r * W + i]; s1 += partial[(long)(r + 1) * W + i];
    s2 += partial[(long)(r + 2) * W + i]; s3 += partial[(long)(r + 3) * W + i];
  }
  for (; r < nrows; ++r) s0 += partial[(long)r * W + i];
  out[i] += (s0 + s1) + (s2 + s3);
}

// ---- thin Linear: Y[n, m] = sum_k X[n, k] W[m, k] + bias[m] for K = 16 (the first conv block: 9 taps of ONE input channel,
// padded to 16 columns) - 16 MACs per output are VALU work next to the 128 bytes a row writes; the MFMA GEMM's tile machinery
// spends 790 us on the 655 MB output at B = 128 x 20 s.  Thread = 8 output channels of one row, weights in registers.
__global__ __launch_bounds__(256) void linear_k16_kernel(const uint4* __restrict__ X, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ bias, uint4* __restrict__ Y, long N_, int M) {
  const int cgs = M >> 3;                                        // channel groups per row (divides 256)
  const long gid = blockIdx.x * 256L + threadIdx.x;
  const int cg = (int)(gid % cgs);
  const long stride = (long)gridDim.x * 256 / cgs;
  float w[8][16], bs[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    bs[m] = bias ? bias[cg * 8 + m] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) w[m][k] = to_f32(W[(long)(cg * 8 + m) * 16 + k]);
  }
  for (long row = gid / cgs; row < N_; row += stride) {
    const uint4 a = X[row * 2], b = X[row * 2 + 1];
    const uint32_t xw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float x[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[2 * k] = bf16_bits_to_f32(xw[k] & 0xffffu); x[2 * k + 1] = bf16_bits_to_f32(xw[k] >> 16); }
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = bs[m];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[m] += w[m][k] * x[k];
    }
    Y[row * cgs + cg] = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                                   pack_bf16x2(acc[6], acc[7]));
  }
}

// ---- direct input gradient of the 3x3 / stride 2 / reflect-pad-1 convolution (C = 64 input, O = 32 output channels) -------
// dx[b,t,f,c] = sum over the output pixels (t2,f2) and taps (dt,df) whose (reflected) source is (t,f) of
//               sum_o dy[b,t2,f2,o] W[o,dt,df,c]
// Replaces the dgrad GEMM (dcol = dy W: 1.47 GB written at B = 128 x 20 s) + col2im (1.47 GB read, element-wise gather):
// 2.1 + 3.2 ms -> one pass that reads dy (82 MB, cache resident) and writes dx (655 MB) once.
// One wave owns 32 consecutive input pixels (one MFMA column each): for every tap whose source exists for at least one of
// them (wave-uniform ballot; stride 2 makes a tap valid for one parity of t / f only, the two reflected edge taps are
// extra virtual taps) the 32 output channels of the source pixel are the B operand, read straight from global memory
// (16 bytes per lane, rows without a source read zeros through the buffer range check); W^T fragments sit in LDS in
// fragment order.  D[channel][pixel]: a lane ends with 4 consecutive channels of its pixel per 8-channel group.
typedef __attribute__((ext_vector_type(8))) __bf16 fe_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fe_f32x16;
typedef uint32_t fe_u32x4 __attribute__((ext_vector_type(4)));

// source coordinate of virtual tap v (0..2: dt = v direct; 3: the mirrored row -1, dt = 0; 4: the mirrored row L, dt = 2)
__device__ __forceinline__ bool conv_s2_src(int v, int t, int L, int L2, int& t2, int& dt) {
  if (v < 3) { dt = v; const int s = t + 1 - v; t2 = s >> 1; return s >= 0 && !(s & 1) && t2 < L2; }
  if (v == 3) { dt = 0; t2 = 0; return t == 1; }
  dt = 2; t2 = (L - 1) >> 1;                                    // 2 t2 + 1 == L: only for odd L
  return (L & 1) && t == L - 2 && t2 < L2;
}

__global__ __launch_bounds__(256) void conv2d_s2_dgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wg,
                                                              bf16_t* __restrict__ dx, int B, int T_, int F, int T2, int F2,
                                                              int Kp, long ntiles) {
  constexpr int C = 64, O = 32;
  __shared__ fe_u32x4 Wl[9 * 2 * 2 * 64];                       // [tap][channel block][o chunk][lane]: 36 KB
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
  for (int e = threadIdx.x; e < 9 * 2 * 2 * 64; e += 256) {
    const int ln = e & 63, oc = (e >> 6) & 1, cb = (e >> 7) & 1, tap = e >> 8;
    const int c = cb * 32 + (ln & 31), o0 = oc * 16 + (ln >> 5) * 8;
    uint32_t w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t lo = reinterpret_cast<const uint16_t*>(wg)[(long)(o0 + 2 * q) * Kp + tap * C + c];
      const uint32_t hi16 = reinterpret_cast<const uint16_t*>(wg)[(long)(o0 + 2 * q + 1) * Kp + tap * C + c];
      w4[q] = lo | (hi16 << 16);
    }
    Wl[e] = fe_u32x4{w4[0], w4[1], w4[2], w4[3]};
  }
  __syncthreads();
  const long npix = (long)B * T_ * F;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), (short)0,
                                                                       (int)((long)B * T2 * F2 * O * 2), 0x00020000);
  for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long pix = tile * 32 + l31;
    const bool pv = pix < npix;
    const long pc = pv ? pix : npix - 1;
    const int f = (int)(pc % F);
    const long r = pc / F;
    const int t = (int)(r % T_), b = (int)(r / T_);
    fe_f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
#pragma unroll 1
    for (int vt = 0; vt < 5; ++vt) {
      int t2, dt;
      const bool srct = conv_s2_src(vt, t, T_, T2, t2, dt);     // (no short-circuit: dt / df are functions of the virtual tap
      const bool okt = pv && srct;                              //  alone and must be set in every lane)
      if (!__builtin_amdgcn_ballot_w64(okt)) continue;
      dt = vt < 3 ? vt : (vt == 3 ? 0 : 2);
      // the three direct f taps together: 6 loads in flight, then 12 MFMAs (a tap that is invalid for a lane reads zeros; the
      // matrix pipe is idle anyway) - one round trip per vt instead of one per tap
      {
        fe_u32x4 bb[3][2];
#pragma unroll
        for (int vf = 0; vf < 3; ++vf) {
          int f2, df;
          const bool ok = okt && conv_s2_src(vf, f, F, F2, f2, df);
          const unsigned off = ok ? (unsigned)((((long)b * T2 + t2) * F2 + f2) * (O * 2) + hi * 16) : 0x80000000u;
          bb[vf][0] = __builtin_bit_cast(fe_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0));
          bb[vf][1] = __builtin_bit_cast(fe_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off + 32, 0, 0));
        }
#pragma unroll
        for (int vf = 0; vf < 3; ++vf) {
          const fe_u32x4* wp = Wl + (dt * 3 + vf) * 256 + lane;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fe_bf16x8, wp[cb * 128]), __builtin_bit_cast(fe_bf16x8, bb[vf][0]), acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fe_bf16x8, wp[cb * 128 + 64]), __builtin_bit_cast(fe_bf16x8, bb[vf][1]), acc[cb], 0, 0, 0);
          }
        }
      }
#pragma unroll 1
      for (int vf = 3; vf < 5; ++vf) {                         // the two mirrored edge columns: rare, skipped by ballot
        int f2, df;
        const bool srcf = conv_s2_src(vf, f, F, F2, f2, df);
        const bool ok = okt && srcf;
        if (!__builtin_amdgcn_ballot_w64(ok)) continue;
        df = vf == 3 ? 0 : 2;
        const unsigned off = ok ? (unsigned)((((long)b * T2 + t2) * F2 + f2) * (O * 2) + hi * 16) : 0x80000000u;
        const fe_u32x4 b0 = __builtin_bit_cast(fe_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0));
        const fe_u32x4 b1 = __builtin_bit_cast(fe_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off + 32, 0, 0));
        const fe_u32x4* wp = Wl + (dt * 3 + df) * 256 + lane;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fe_bf16x8, wp[cb * 128]), __builtin_bit_cast(fe_bf16x8, b0), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fe_bf16x8, wp[cb * 128 + 64]), __builtin_bit_cast(fe_bf16x8, b1), acc[cb], 0, 0, 0);
        }
      }
    }
    if (pv) {
      bf16_t* o = dx + pix * C + hi * 4;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(o + cb * 32 + g * 8) =
              make_uint2(pack_bf16x2(acc[cb][g * 4], acc[cb][g * 4 + 1]), pack_bf16x2(acc[cb][g * 4 + 2], acc[cb][g * 4 + 3]));
    }
  }
}

// ---- InputNormalization (speechbrain.processing.features.InputNormalization, recipe key `normalize`) ---------------
// per-utterance mean and unbiased std over the valid frames t < len[b] of every feature; grid (ceil(F/16), B), 256
// threads = 16 features x 16 time groups (four loads in flight per thread); two passes over the (small) feature block for a
// stable variance.  (64 features x 4 time groups walked 375-500 dependent steps per thread on 2 workgroups per utterance:
// 83 us for the recipe batch's 10 x 1501 frames, the largest launch of its feature pipeline.)
template <typename T>
__global__ __launch_bounds__(256) void utt_meanstd_kernel(const T* __restrict__ X, long ldx, int Tmax, int F,
                                                          const int* __restrict__ len, float* __restrict__ mean,
                                                          float* __restrict__ sd, int mean_norm, int std_norm, float eps) {
  constexpr int CF = 16, TG = 16;
  __shared__ float red[TG][CF];
  const int cl = threadIdx.x & (CF - 1), tg = threadIdx.x / CF, b = blockIdx.y;
  const int c = blockIdx.x * CF + cl;
  const int n = min(max(len[b], 0), Tmax);
  const T* x = X + (long)b * Tmax * ldx + (c < F ? c : 0);
  auto fold = [&]() {                                    // the 16 time groups in a fixed order
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < TG; ++g) v += red[g][cl];
    return v;
  };
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = tg;
  for (; t + 3 * TG < n; t += 4 * TG) {
    s0 += to_f32(x[(long)t * ldx]); s1 += to_f32(x[(long)(t + TG) * ldx]);
    s2 += to_f32(x[(long)(t + 2 * TG) * ldx]); s3 += to_f32(x[(long)(t + 3 * TG) * ldx]);
  }
  for (; t < n; t += TG) s0 += to_f32(x[(long)t * ldx]);
  red[tg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  const float m = n > 0 ? fold() / (float)n : 0.f;
  __syncthreads();
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  t = tg;
  for (; t + 3 * TG < n; t += 4 * TG) {
    const float d0 = to_f32(x[(long)t * ldx]) - m, d1 = to_f32(x[(long)(t + TG) * ldx]) - m;
    const float d2 = to_f32(x[(long)(t + 2 * TG) * ldx]) - m, d3 = to_f32(x[(long)(t + 3 * TG) * ldx]) - m;
    q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
  }
  for (; t < n; t += TG) { const float d = to_f32(x[(long)t * ldx]) - m; q0 += d * d; }
  red[tg][cl] = (q0 + q1) + (q2 + q3);
  __syncthreads();
  if (tg == 0 && c < F) {
    const float var = n > 1 ? fold() / (float)(n - 1) : __builtin_nanf("");
    mean[(long)b * F + c] = mean_norm ? m : 0.f;
    sd[(long)b * F + c] = std_norm ? fmaxf(sqrtf(var), eps) : 1.f;    // torch.max(std, eps): NaN (one frame) propagates
  }
}

// glob = (1 - w) * glob + w * mean_b(cur)   (w = 1: replace).  One thread per feature, fixed order over the batch.
__global__ void stats_combine_kernel(const float* __restrict__ cur_mean, const float* __restrict__ cur_std, int B, int F,
                                     float* glob_mean, float* glob_std, float w) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= F) return;
  float m = 0.f, s = 0.f;
  for (int b = 0; b < B; ++b) { m += cur_mean[(long)b * F + c]; s += cur_std[(long)b * F + c]; }
  m /= (float)B; s /= (float)B;
  glob_mean[c] = w >= 1.f ? m : (1.f - w) * glob_mean[c] + w * m;
  glob_std[c] = w >= 1.f ? s : (1.f - w) * glob_std[c] + w * s;
}

// Y[b,t,c] = (X[b,t,c] - mean[b*stride + c]) / std[b*stride + c]   (stride 0: shared statistics)
template <typename T>
__global__ __launch_bounds__(256) void colnorm_kernel(const T* __restrict__ X, long ldx, const float* __restrict__ mean,
                                                      const float* __restrict__ sd, long sstride, T* __restrict__ Y, long ldy,
                                                      int Tmax, int F, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % F);
    const long row = i / F;
    const long b = row / Tmax;
    Y[row * ldy + c] = from_f32<T>((to_f32(X[row * ldx + c]) - mean[b * sstride + c]) / sd[b * sstride + c]);
  }
}

static inline int fgrid(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace smx

using namespace smx;
#define STREAM reinterpret_cast<hipStream_t>(stream)

extern "C" int smx_frame_window(const float* wav, int64_t ldw, const float* window, float* frames, int B, int L, int T,
                                int n_fft, int hop, void* stream) {
  SMX_REQUIRE(wav && window && frames && n_fft % 4 == 0 && hop > 0, "smx_frame_window: bad arguments (n_fft % 4 == 0)");
  if (B <= 0 || T <= 0) return SMX_OK;
  hipLaunchKernelGGL(frame_window_kernel, dim3(fgrid((long)B * T * (n_fft / 4))), dim3(256), 0, STREAM, wav, ldw, window, frames,
                     B, L, T, n_fft, hop);
  return check_launch("smx_frame_window");
}

extern "C" size_t smx_fbank_workspace(int B, int T, int n_mels) {
  return ((size_t)B * T * n_mels + (size_t)B * T + B + 16) * sizeof(float);   // dB values | one maximum per frame | one per utterance
}

extern "C" int smx_mel_db(int out_dtype, const float* spec, int64_t lds, int im_off, const float* fb, int n_bins, int n_mels,
                          float amin, float top_db, void* out, int B, int T, void* workspace, void* stream) {
  SMX_REQUIRE(spec && fb && out && workspace && n_bins > 0 && n_mels > 0, "smx_mel_db: bad arguments");
  if (B <= 0 || T <= 0) return SMX_OK;
  const int N = B * T;
  float* db = reinterpret_cast<float*>(workspace);
  float* bmax = db + (size_t)N * n_mels;
  float* umax = bmax + N;
  SMX_REQUIRE(n_mels <= 256, "smx_mel_db: n_mels=%d > 256", n_mels);
  int mblocks = (N + 3) / 4;
  if (mblocks > 2048) mblocks = 2048;
  hipLaunchKernelGGL(mel_db_kernel, dim3(mblocks), dim3(256), (4 * n_bins + n_mels * 48) * sizeof(float), STREAM, spec, lds, im_off, fb, n_bins,
                     n_mels, amin, db, bmax, N);
  hipLaunchKernelGGL(utt_max_kernel, dim3(B), dim3(256), 0, STREAM, bmax, T, 1, umax);
  const long total = (long)N * n_mels;
  if (out_dtype == SMX_BF16) hipLaunchKernelGGL((topdb_clamp_kernel<bf16_t>), dim3(fgrid(total)), dim3(256), 0, STREAM, db, umax, top_db, (bf16_t*)out, (long)T * n_mels, total);
  else hipLaunchKernelGGL((topdb_clamp_kernel<float>), dim3(fgrid(total)), dim3(256), 0, STREAM, db, umax, top_db, (float*)out, (long)T * n_mels, total);
  return check_launch("smx_mel_db");
}

extern "C" int smx_im2col_s2(int dtype, const void* x, void* col, int B, int T, int F, int C, int Kp, void* stream) {
  SMX_REQUIRE(x && col && Kp >= 9 * C && T >= 2 && F >= 2, "smx_im2col_s2: bad arguments");
  const int T2 = (T + 1) / 2, F2 = (F + 1) / 2;
  const long total = (long)B * T2 * F2 * Kp;
  if (total <= 0) return SMX_OK;
  if (dtype == SMX_BF16 && C % 8 == 0 && Kp == 9 * C && aligned16(x) && aligned16(col)) {
    const long rows = (long)B * T2 * F2;
    long g = (rows + 3) / 4;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(im2col_s2_vec_kernel, dim3((unsigned)g), dim3(256), 0, STREAM, (const uint4*)x, (uint4*)col, B, T, F, C / 8, T2, F2, Kp / 8);
    return check_launch("smx_im2col_s2");
  }
  if (dtype == SMX_BF16) hipLaunchKernelGGL((im2col_s2_kernel<bf16_t>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const bf16_t*)x, (bf16_t*)col, B, T, F, C, T2, F2, Kp);
  else hipLaunchKernelGGL((im2col_s2_kernel<float>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const float*)x, (float*)col, B, T, F, C, T2, F2, Kp);
  return check_launch("smx_im2col_s2");
}

static int conv1_blocks(long rows) { return (int)(rows < 768 ? rows : 768); }   // persistent: 3 workgroups per CU, 768 partial rows
static bool conv1_shape_ok(int T, int F, int O) { return O == 64 && F % 16 == 0 && F >= 16 && F <= 160 && (F / 2) / 8 <= 10 && T >= 2; }
extern "C" size_t smx_conv1_ln_workspace(int B, int T, int F, int O) {
  if (B <= 0 || T <= 0 || !conv1_shape_ok(T, F, O)) return 0;
  const long rows = (long)B * ((T + 1) / 2);
  return (size_t)conv1_blocks(rows) * (2 * (F / 2) * O + O * 10) * sizeof(float);
}
#define SMX_CONV1_NI(NI_, CALL)                                           \
  switch (NI_) {                                                          \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break; \
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; case 9: CALL(9); break; default: CALL(10); break; \
  }
extern "C" int smx_conv1_ln_fwd(int dtype, const void* X, const float* W9, const float* bias, const float* gamma, const float* beta,
                                float eps, int act, void* Y, float* stats, int B, int T, int F, int O, void* stream) {
  SMX_REQUIRE(X && W9 && gamma && beta && Y, "smx_conv1_ln_fwd: null pointer");
  if (!conv1_shape_ok(T, F, O)) return fail(SMX_EUNSUPPORTED, "smx_conv1_ln_fwd: built for O = 64, F a multiple of 16 up to 160");
  if (B <= 0) return SMX_OK;
  const int T2 = (T + 1) / 2, NI = (F / 2) / 8;
  const long rows = (long)B * T2;
  long g = rows < 256L * 16 ? rows : 256L * 16;
#define CALL(N_)                                                                                                              \
  do {                                                                                                                         \
    if (dtype == SMX_BF16) hipLaunchKernelGGL((conv1_ln_fwd_kernel<bf16_t, N_>), dim3((unsigned)g), dim3(256), 0, STREAM, (const bf16_t*)X, W9, bias, gamma, beta, eps, act, (bf16_t*)Y, stats, B, T, F, T2); \
    else hipLaunchKernelGGL((conv1_ln_fwd_kernel<float, N_>), dim3((unsigned)g), dim3(256), 0, STREAM, (const float*)X, W9, bias, gamma, beta, eps, act, (float*)Y, stats, B, T, F, T2); \
  } while (0)
  SMX_CONV1_NI(NI, CALL)
#undef CALL
  return check_launch("smx_conv1_ln_fwd");
}
extern "C" int smx_conv1_ln_bwd(int dtype, const void* dA, const void* X, const float* W9, const float* bias, const float* gamma,
                                const float* beta, const float* stats, int act, float* grads, void* workspace, int B, int T, int F,
                                int O, void* stream) {
  SMX_REQUIRE(dA && X && W9 && gamma && beta && stats && grads && workspace, "smx_conv1_ln_bwd: null pointer");
  if (!conv1_shape_ok(T, F, O)) return fail(SMX_EUNSUPPORTED, "smx_conv1_ln_bwd: built for O = 64, F a multiple of 16 up to 160");
  if (B <= 0) return SMX_OK;
  const int T2 = (T + 1) / 2, NI = (F / 2) / 8;
  const long rows = (long)B * T2;
  const int g = conv1_blocks(rows);
  float* partial = reinterpret_cast<float*>(workspace);
#define CALL(N_)                                                                                                              \
  do {                                                                                                                         \
    if (dtype == SMX_BF16) hipLaunchKernelGGL((conv1_ln_bwd_kernel<bf16_t, N_>), dim3(g), dim3(256), 0, STREAM, (const bf16_t*)dA, (const bf16_t*)X, W9, bias, gamma, beta, stats, act, partial, B, T, F, T2); \
    else hipLaunchKernelGGL((conv1_ln_bwd_kernel<float, N_>), dim3(g), dim3(256), 0, STREAM, (const float*)dA, (const float*)X, W9, bias, gamma, beta, stats, act, partial, B, T, F, T2); \
  } while (0)
  SMX_CONV1_NI(NI, CALL)
#undef CALL
  const long W = 2L * (F / 2) * O + O * 10;
  hipLaunchKernelGGL(rows_sum_add_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, STREAM, partial, g, W, grads);
  return check_launch("smx_conv1_ln_bwd");
}

extern "C" int smx_linear_k16_fwd(int dtype, const void* X, const void* W, const float* bias, void* Y, int64_t N, int M, void* stream) {
  SMX_REQUIRE(X && W && Y, "smx_linear_k16_fwd: null pointer");
  if (dtype != SMX_BF16 || M % 8 != 0 || 256 % (M / 8) != 0 || !aligned16(X) || !aligned16(Y))
    return fail(SMX_EUNSUPPORTED, "smx_linear_k16_fwd: bf16, M a multiple of 8 with M / 8 dividing 256, 16-byte aligned rows");
  if (N <= 0) return SMX_OK;
  long g = (N * (M / 8) + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(linear_k16_kernel, dim3((unsigned)g), dim3(256), 0, STREAM, (const uint4*)X, (const bf16_t*)W, bias, (uint4*)Y, (long)N, M);
  return check_launch("smx_linear_k16_fwd");
}

extern "C" int smx_conv2d_s2_dgrad(int dtype, const void* dY, const void* Wg, void* dX, int B, int T, int F, int C, int O, int Kp,
                                   void* stream) {
  SMX_REQUIRE(dY && Wg && dX && T >= 4 && F >= 4 && Kp >= 9 * C, "smx_conv2d_s2_dgrad: bad arguments");
  if (dtype != SMX_BF16 || C != 64 || O != 32) return fail(SMX_EUNSUPPORTED, "smx_conv2d_s2_dgrad: built for bf16, C = 64, O = 32");
  const int T2 = (T + 1) / 2, F2 = (F + 1) / 2;
  SMX_REQUIRE((long)B * T2 * F2 * O * 2 < (1L << 31), "smx_conv2d_s2_dgrad: dY >= 2 GB");
  SMX_REQUIRE(aligned16(dY) && aligned16(dX), "smx_conv2d_s2_dgrad: 16-byte aligned tensors");
  const long ntiles = ((long)B * T * F + 31) / 32;
  long g = (ntiles + 3) / 4;
  if (g > 256 * 4) g = 256 * 4;                            // persistent workgroups: the 36 KB weight image is built once each
  hipLaunchKernelGGL(conv2d_s2_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, STREAM, (const bf16_t*)dY, (const bf16_t*)Wg,
                     (bf16_t*)dX, B, T, F, T2, F2, Kp, ntiles);
  return check_launch("smx_conv2d_s2_dgrad");
}

extern "C" int smx_col2im_s2(int dtype, const void* dcol, void* dx, int B, int T, int F, int C, int Kp, void* stream) {
  SMX_REQUIRE(dcol && dx && Kp >= 9 * C && T >= 2 && F >= 2, "smx_col2im_s2: bad arguments");
  const int T2 = (T + 1) / 2, F2 = (F + 1) / 2;
  const long total = (long)B * T * F * C;
  if (total <= 0) return SMX_OK;
  if (dtype == SMX_BF16) hipLaunchKernelGGL((col2im_s2_kernel<bf16_t>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const bf16_t*)dcol, (bf16_t*)dx, B, T, F, C, T2, F2, Kp);
  else hipLaunchKernelGGL((col2im_s2_kernel<float>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const float*)dcol, (float*)dx, B, T, F, C, T2, F2, Kp);
  return check_launch("smx_col2im_s2");
}

extern "C" int smx_utt_meanstd(int dtype, const void* X, int64_t ldx, const int32_t* len, float* mean, float* std, int B, int T,
                               int F, int mean_norm, int std_norm, float eps, void* stream) {
  SMX_REQUIRE(X && len && mean && std && T > 0 && F > 0, "smx_utt_meanstd: bad arguments");
  if (B <= 0) return SMX_OK;
  dim3 grid((F + 15) / 16, B);
  if (dtype == SMX_BF16) hipLaunchKernelGGL((utt_meanstd_kernel<bf16_t>), grid, dim3(256), 0, STREAM, (const bf16_t*)X, ldx, T, F, len, mean, std, mean_norm, std_norm, eps);
  else hipLaunchKernelGGL((utt_meanstd_kernel<float>), grid, dim3(256), 0, STREAM, (const float*)X, ldx, T, F, len, mean, std, mean_norm, std_norm, eps);
  return check_launch("smx_utt_meanstd");
}

extern "C" int smx_stats_combine(const float* cur_mean, const float* cur_std, int B, int F, float* glob_mean, float* glob_std,
                                 float weight, void* stream) {
  SMX_REQUIRE(cur_mean && cur_std && glob_mean && glob_std && B > 0 && F > 0 && weight > 0.f, "smx_stats_combine: bad arguments");
  hipLaunchKernelGGL(stats_combine_kernel, dim3((F + 255) / 256), dim3(256), 0, STREAM, cur_mean, cur_std, B, F, glob_mean,
                     glob_std, weight);
  return check_launch("smx_stats_combine");
}

extern "C" int smx_colnorm(int dtype, const void* X, int64_t ldx, const float* mean, const float* std, int64_t stat_stride,
                           void* Y, int64_t ldy, int B, int T, int F, void* stream) {
  SMX_REQUIRE(X && mean && std && Y && T > 0 && F > 0, "smx_colnorm: bad arguments");
  if (B <= 0) return SMX_OK;
  const long total = (long)B * T * F;
  if (dtype == SMX_BF16) hipLaunchKernelGGL((colnorm_kernel<bf16_t>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const bf16_t*)X, ldx, mean, std, stat_stride, (bf16_t*)Y, ldy, T, F, total);
  else hipLaunchKernelGGL((colnorm_kernel<float>), dim3(fgrid(total)), dim3(256), 0, STREAM, (const float*)X, ldx, mean, std, stat_stride, (float*)Y, ldy, T, F, total);
  return check_launch("smx_colnorm");
}
