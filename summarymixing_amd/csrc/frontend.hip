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
  int lo[MPL], hi[MPL];
#pragma unroll
  for (int k = 0; k < MPL; ++k) {
    const int m = lane + 64 * k;
    lo[k] = n_bins; hi[k] = 0;
    if (m < n_mels) {
      const float* fr = fb + (long)m * n_bins;
      for (int f = 0; f < n_bins; ++f)
        if (fr[f] != 0.f) { lo[k] = min(lo[k], f); hi[k] = f + 1; }
      // the band's weights are the same for every frame: once into LDS (a global load per tap and frame made this kernel
      // latency-bound: 845 us for 256 000 frames)
      if (w == 0)
        for (int j = 0; j < MELW; ++j) band[m * MELW + j] = (lo[k] + j < hi[k]) ? fr[lo[k] + j] : 0.f;
    }
  }
  __syncthreads();
  for (int n = blockIdx.x * 4 + w; n < N_; n += gridDim.x * 4) {
    const float* s = S + (long)n * lds;
    for (int f = lane; f < n_bins; f += 64) { const float re = s[f], im = s[im_off + f]; P[f] = re * re + im * im; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes are visible to all its lanes
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
        db[(long)n * n_mels + m] = 10.f * log10f(fmaxf(a0 + a1, amin));
      }
    }
    __builtin_amdgcn_wave_barrier();                     // (P is rewritten by the next frame)
  }
  (void)bmax;
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
// per-utterance mean and unbiased std over the valid frames t < len[b] of every feature; grid (ceil(F/64), B), 256
// threads = 64 features x 4 time groups; two passes over the (small) feature block for a stable variance
template <typename T>
__global__ __launch_bounds__(256) void utt_meanstd_kernel(const T* __restrict__ X, long ldx, int Tmax, int F,
                                                          const int* __restrict__ len, float* __restrict__ mean,
                                                          float* __restrict__ sd, int mean_norm, int std_norm, float eps) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6, b = blockIdx.y;
  const int c = blockIdx.x * 64 + cl;
  const int n = min(max(len[b], 0), Tmax);
  const T* x = X + (long)b * Tmax * ldx + c;
  float s = 0.f;
  if (c < F) for (int t = tg; t < n; t += 4) s += to_f32(x[(long)t * ldx]);
  red[tg][cl] = s;
  __syncthreads();
  const float m = n > 0 ? ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)n : 0.f;
  __syncthreads();
  float q = 0.f;
  if (c < F) for (int t = tg; t < n; t += 4) { const float d = to_f32(x[(long)t * ldx]) - m; q += d * d; }
  red[tg][cl] = q;
  __syncthreads();
  if (tg == 0 && c < F) {
    const float var = n > 1 ? ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)(n - 1) : __builtin_nanf("");
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
  return ((size_t)B * T * n_mels + (size_t)(B * (long)T + 3) / 4 + B + 16) * sizeof(float);
}

extern "C" int smx_mel_db(int out_dtype, const float* spec, int64_t lds, int im_off, const float* fb, int n_bins, int n_mels,
                          float amin, float top_db, void* out, int B, int T, void* workspace, void* stream) {
  SMX_REQUIRE(spec && fb && out && workspace && n_bins > 0 && n_mels > 0, "smx_mel_db: bad arguments");
  if (B <= 0 || T <= 0) return SMX_OK;
  const int N = B * T;
  float* db = reinterpret_cast<float*>(workspace);
  float* bmax = db + (size_t)N * n_mels;
  float* umax = bmax + (N + 3) / 4;
  SMX_REQUIRE(n_mels <= 256, "smx_mel_db: n_mels=%d > 256", n_mels);
  int mblocks = (N + 3) / 4;
  if (mblocks > 2048) mblocks = 2048;
  hipLaunchKernelGGL(mel_db_kernel, dim3(mblocks), dim3(256), (4 * n_bins + n_mels * 48) * sizeof(float), STREAM, spec, lds, im_off, fb, n_bins,
                     n_mels, amin, db, bmax, N);
  hipLaunchKernelGGL(utt_max_kernel, dim3(B), dim3(256), 0, STREAM, db, T, n_mels, umax);
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
  dim3 grid((F + 63) / 64, B);
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
