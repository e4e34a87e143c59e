// ctc.hip — the CTC head that follows the encoder in the reference recipes, gfx950 only:
//   log-softmax over the vocabulary (speechbrain.nnet.activations.Softmax(apply_log=True), recipe `log_softmax`) and
//   the CTC negative log-likelihood + its gradient (speechbrain.nnet.losses.ctc_loss -> torch.nn.functional.ctc_loss
//   with zero_infinity=True; recipe `ctc_cost`, …transducer.yaml:297-298,331).
// Three kernels for the loss: alpha (forward variable, one workgroup per utterance, states across the threads, one
// LDS-resident time step at a time), beta (backward variable; turns the stored alphas into log occupancies
// log(alpha*beta/y) in place) and a fully parallel gradient pass (one workgroup per (utterance, frame) row).
#include "smx_common.h"

namespace smx {

static constexpr float NEG_INF = -__builtin_inff();

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// ---- log-softmax over the last dimension: one wave per row, row read twice (max+sum, then write) -------------
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                              int N_, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N_) return;
  const T* x = X + (long)row * ldx;
  float m = NEG_INF;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, to_f32(x[c]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += expf(to_f32(x[c]) - m);
  s = wave_sum(s);
  const float lz = m + logf(s);
  T* y = Y + (long)row * ldy;
  for (int c = lane; c < V; c += 64) y[c] = from_f32<T>(to_f32(x[c]) - lz);
}

// dX = dY - exp(Y) * sum_v dY   (Y = the log-probabilities the forward produced)
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ Y,
                                                              long ldy, T* __restrict__ dX, long lddx, int N_, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N_) return;
  const T* dy = dY + (long)row * lddy;
  const T* y = Y + (long)row * ldy;
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += to_f32(dy[c]);
  s = wave_sum(s);
  T* dx = dX + (long)row * lddx;
  for (int c = lane; c < V; c += 64) dx[c] = from_f32<T>(to_f32(dy[c]) - expf(to_f32(y[c])) * s);
}

// ---- CTC ---------------------------------------------------------------------------------------------------
// extended label sequence l' of utterance b: state s is the blank for even s, target (s >> 1) for odd s
__device__ __forceinline__ int state_label(const int* tg, int s, int blank) { return (s & 1) ? tg[s >> 1] : blank; }

// alpha[b, t, s] (log domain, includes the emission at t); nll[b] = -log p(l | x).  grid = B, block = 256.
// A thread owns the states s = tid + 256 k (k < KS): their labels and skip permissions live in registers for the whole
// utterance and the emissions of frame t+1 are requested before frame t is combined, so the gather latency is off the
// T-step critical path (one barrier + one LDS exchange per frame remain).
template <typename T, int KS>
__global__ __launch_bounds__(256) void ctc_alpha_kernel(const T* __restrict__ LP, long ldlp, int Tmax, int V,
                                                        const int* __restrict__ targets, int Smax,
                                                        const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                                        int blank, float* __restrict__ alpha, int Lmax, float* __restrict__ nll) {
  extern __shared__ float sh[];                          // 2 * Lmax floats
  const int b = blockIdx.x;
  const int Tb = min(in_len[b], Tmax), S = min(tgt_len[b], Smax), L = 2 * S + 1;
  const int* tg = targets + (long)b * Smax;
  float* prev = sh;
  float* cur = sh + Lmax;
  float* ab = alpha + (long)b * Tmax * Lmax;
  int lab[KS];
  bool skip[KS];
  float y[KS], yn[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int s = threadIdx.x + 256 * k;
    lab[k] = s < L ? state_label(tg, s, blank) : blank;
    skip[k] = s < L && s >= 2 && (s & 1) && tg[s >> 1] != tg[(s >> 1) - 1];
    y[k] = yn[k] = 0.f;
  }
  const T* lp0 = LP + (long)b * Tmax * ldlp;
  if (Tb > 0) {
#pragma unroll
    for (int k = 0; k < KS; ++k) y[k] = to_f32(lp0[lab[k]]);
  }
  for (int t = 0; t < Tb; ++t) {
    if (t + 1 < Tb) {
      const T* lpn = lp0 + (long)(t + 1) * ldlp;
#pragma unroll
      for (int k = 0; k < KS; ++k) yn[k] = to_f32(lpn[lab[k]]);
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int s = threadIdx.x + 256 * k;
      if (s < L) {
        float a;
        if (t == 0) {
          a = s < 2 ? y[k] : NEG_INF;
        } else {
          const float a0 = prev[s];
          const float a1 = s >= 1 ? prev[s - 1] : NEG_INF;
          const float a2 = skip[k] ? prev[s - 2] : NEG_INF;
          a = lse3(a0, a1, a2) + y[k];
        }
        cur[s] = a;
        ab[(long)t * Lmax + s] = a;
      }
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
#pragma unroll
    for (int k = 0; k < KS; ++k) y[k] = yn[k];
  }
  if (threadIdx.x == 0) {
    float ll = NEG_INF;
    if (Tb > 0) ll = lse3(prev[L - 1], L > 1 ? prev[L - 2] : NEG_INF, NEG_INF);
    nll[b] = -ll;                                        // +inf when no alignment exists (zero_infinity handled by the caller)
  }
}

// beta recursion; overwrites alpha[b,t,s] with the log occupancy  alpha + beta - y  ( = log(alpha*beta / y) ).
template <typename T, int KS>
__global__ __launch_bounds__(256) void ctc_beta_kernel(const T* __restrict__ LP, long ldlp, int Tmax, int V,
                                                       const int* __restrict__ targets, int Smax,
                                                       const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                                       int blank, float* __restrict__ alpha, int Lmax) {
  extern __shared__ float sh[];
  const int b = blockIdx.x;
  const int Tb = min(in_len[b], Tmax), S = min(tgt_len[b], Smax), L = 2 * S + 1;
  const int* tg = targets + (long)b * Smax;
  float* nxt = sh;
  float* cur = sh + Lmax;
  float* ab = alpha + (long)b * Tmax * Lmax;
  int lab[KS];
  bool skip[KS];
  float y[KS], yn[KS], av[KS], avn[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int s = threadIdx.x + 256 * k;
    lab[k] = s < L ? state_label(tg, s, blank) : blank;
    skip[k] = s + 2 < L && (s & 1) && tg[s >> 1] != tg[(s >> 1) + 1];
    y[k] = yn[k] = av[k] = avn[k] = 0.f;
  }
  const T* lp0 = LP + (long)b * Tmax * ldlp;
  if (Tb > 0) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int s = threadIdx.x + 256 * k;
      y[k] = to_f32(lp0[(long)(Tb - 1) * ldlp + lab[k]]);
      av[k] = s < L ? ab[(long)(Tb - 1) * Lmax + s] : NEG_INF;
    }
  }
  for (int t = Tb - 1; t >= 0; --t) {
    if (t > 0) {                                         // frame t-1: emissions and alphas, one step ahead
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int s = threadIdx.x + 256 * k;
        yn[k] = to_f32(lp0[(long)(t - 1) * ldlp + lab[k]]);
        avn[k] = s < L ? ab[(long)(t - 1) * Lmax + s] : NEG_INF;
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int s = threadIdx.x + 256 * k;
      if (s < L) {
        float bt;
        if (t == Tb - 1) {
          bt = s >= L - 2 ? y[k] : NEG_INF;
        } else {
          const float b0 = nxt[s];
          const float b1 = s + 1 < L ? nxt[s + 1] : NEG_INF;
          const float b2 = skip[k] ? nxt[s + 2] : NEG_INF;
          bt = lse3(b0, b1, b2) + y[k];
        }
        cur[s] = bt;
        ab[(long)t * Lmax + s] = (av[k] == NEG_INF || bt == NEG_INF) ? NEG_INF : av[k] + bt - y[k];
      }
    }
    __syncthreads();
    float* tmp = nxt; nxt = cur; cur = tmp;
#pragma unroll
    for (int k = 0; k < KS; ++k) { y[k] = yn[k]; av[k] = avn[k]; }
  }
}

// gradient w.r.t. the (log-softmax-normalised) inputs, torch.nn.functional.ctc_loss convention (Graves eq. 16):
//   G[b,t,v] = gscale[b] * ( exp(lp[b,t,v]) - exp( log sum_{s: l'_s = v} occ[b,t,s] + nll[b] ) )
// and 0 for t >= in_len[b] or when nll[b] is infinite (zero_infinity).  grid = B*Tmax rows, block = 256.
// Occurrence chains of the target labels of one utterance (independent of t): first[k] = 1 when no k' < k carries the same
// label, next[k] = the next k' > k with the same label or -1.  The gradient kernel sums the occupancies of a label along its
// chain, ONE thread per label in increasing k: no atomics, bit-reproducible (round 3; LDS float atomics before).
__global__ __launch_bounds__(256) void ctc_chain_kernel(const int* __restrict__ targets, int Smax, const int* __restrict__ tgt_len,
                                                        int* __restrict__ chain) {
  const int b = blockIdx.x, S = min(tgt_len[b], Smax);
  const int* tg = targets + (long)b * Smax;
  int* nxt = chain + (long)b * 2 * Smax;
  int* first = nxt + Smax;
  for (int k = threadIdx.x; k < S; k += 256) {
    const int c = tg[k];
    int n = -1, f = 1;
    for (int j = k + 1; j < S; ++j)
      if (tg[j] == c) { n = j; break; }
    for (int j = 0; j < k; ++j)
      if (tg[j] == c) { f = 0; break; }
    nxt[k] = n;
    first[k] = f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ LP, long ldlp, int Tmax, int V,
                                                       const int* __restrict__ targets, int Smax,
                                                       const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                                       int blank, const float* __restrict__ occ, int Lmax,
                                                       const float* __restrict__ nll, const float* __restrict__ gscale,
                                                       const int* __restrict__ chain, T* __restrict__ G, long ldg) {
  extern __shared__ float acc[];                         // V floats | Lmax floats (exp(occ - max) per state)
  __shared__ float redm[4];
  __shared__ float redb[4];
  float* es = acc + V;
  const int row = blockIdx.x, b = row / Tmax, t = row % Tmax;
  T* g = G + (long)row * ldg;
  const float nl = nll[b];
  const int Tb = min(in_len[b], Tmax);
  if (t >= Tb || !(nl < __builtin_inff())) {
    for (int v = threadIdx.x; v < V; v += 256) g[v] = from_f32<T>(0.f);
    return;
  }
  const int S = min(tgt_len[b], Smax), L = 2 * S + 1;
  const int* tg = targets + (long)b * Smax;
  const int* nxt = chain + (long)b * 2 * Smax;
  const int* first = nxt + Smax;
  const float* oc = occ + ((long)b * Tmax + t) * Lmax;
  for (int v = threadIdx.x; v < V; v += 256) acc[v] = 0.f;
  float m = NEG_INF;
  for (int s = threadIdx.x; s < L; s += 256) m = fmaxf(m, oc[s]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float bsum = 0.f;
  if (m > NEG_INF) {
    for (int s = threadIdx.x; s < L; s += 256) {
      const float o = oc[s];
      es[s] = o > NEG_INF ? expf(o - m) : 0.f;
    }
  }
  __syncthreads();
  if (m > NEG_INF) {
    // blank states (even s): per-thread sums in increasing s, then a fixed-order fold
    for (int s = 2 * threadIdx.x; s < L; s += 512) bsum += es[s];
    bsum = wave_sum(bsum);
    if ((threadIdx.x & 63) == 0) redb[threadIdx.x >> 6] = bsum;
    // label states (odd s = 2 k + 1): the first occurrence of a label walks its chain
    for (int k = threadIdx.x; k < S; k += 256) {
      if (first[k]) {
        float sum = es[2 * k + 1];
        for (int j = nxt[k]; j >= 0; j = nxt[j]) sum += es[2 * j + 1];
        if (tg[k] != blank) acc[tg[k]] = sum;              // (a label equal to `blank` is invalid input, as for torch's ctc_loss:
                                                           //  it is ignored here instead of racing with the blank sum below)
      }
    }
  }
  __syncthreads();
  if (m > NEG_INF && threadIdx.x == 0) acc[blank] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
  __syncthreads();
  const float sc = gscale[b];
  const T* lp = LP + (long)row * ldlp;
  const float shift = m + nl;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float a = acc[v];
    const float lab = a > 0.f ? expf(logf(a) + shift) : 0.f;
    g[v] = from_f32<T>(sc * (expf(to_f32(lp[v])) - lab));
  }
}

}  // namespace smx

using namespace smx;
#define STREAM reinterpret_cast<hipStream_t>(stream)

extern "C" int smx_log_softmax_fwd(int dtype, const void* X, int64_t ldx, void* Y, int64_t ldy, int N, int V, void* stream) {
  SMX_REQUIRE(X && Y && V > 0, "smx_log_softmax_fwd: bad arguments");
  if (N <= 0) return SMX_OK;
  dim3 grid((N + 3) / 4);
  if (dtype == SMX_BF16) hipLaunchKernelGGL((log_softmax_fwd_kernel<bf16_t>), grid, dim3(256), 0, STREAM, (const bf16_t*)X, ldx, (bf16_t*)Y, ldy, N, V);
  else hipLaunchKernelGGL((log_softmax_fwd_kernel<float>), grid, dim3(256), 0, STREAM, (const float*)X, ldx, (float*)Y, ldy, N, V);
  return check_launch("smx_log_softmax_fwd");
}

extern "C" int smx_log_softmax_bwd(int dtype, const void* dY, int64_t lddy, const void* Y, int64_t ldy, void* dX, int64_t lddx,
                                   int N, int V, void* stream) {
  SMX_REQUIRE(dY && Y && dX && V > 0, "smx_log_softmax_bwd: bad arguments");
  if (N <= 0) return SMX_OK;
  dim3 grid((N + 3) / 4);
  if (dtype == SMX_BF16) hipLaunchKernelGGL((log_softmax_bwd_kernel<bf16_t>), grid, dim3(256), 0, STREAM, (const bf16_t*)dY, lddy, (const bf16_t*)Y, ldy, (bf16_t*)dX, lddx, N, V);
  else hipLaunchKernelGGL((log_softmax_bwd_kernel<float>), grid, dim3(256), 0, STREAM, (const float*)dY, lddy, (const float*)Y, ldy, (float*)dX, lddx, N, V);
  return check_launch("smx_log_softmax_bwd");
}

static int ctc_lmax(int Smax) { return 2 * Smax + 1; }

// [forward / occupancy variables (B, T, Lmax) fp32][label occurrence chains (B, 2, Smax) int32]
static size_t ctc_alpha_bytes(int B, int T, int Smax) { return ((size_t)B * T * ctc_lmax(Smax) * sizeof(float) + 15) / 16 * 16; }
extern "C" size_t smx_ctc_workspace(int B, int T, int Smax) {
  return ctc_alpha_bytes(B, T, Smax) + (size_t)B * 2 * (Smax > 0 ? Smax : 1) * sizeof(int);
}

extern "C" int smx_ctc_loss_fwd(int dtype, const void* log_probs, int64_t ldlp, const int32_t* targets, const int32_t* in_len,
                                const int32_t* tgt_len, int B, int T, int V, int Smax, int blank, float* nll, void* workspace,
                                void* stream) {
  SMX_REQUIRE(log_probs && targets && in_len && tgt_len && nll && workspace, "smx_ctc_loss_fwd: null pointer");
  SMX_REQUIRE(B >= 0 && T > 0 && V > 0 && Smax >= 0 && blank >= 0 && blank < V, "smx_ctc_loss_fwd: bad sizes");
  const int Lmax = ctc_lmax(Smax);
  SMX_REQUIRE(2 * Lmax * sizeof(float) <= 64 * 1024, "smx_ctc_loss_fwd: target length %d too long (max 4095)", Smax);
  if (B == 0) return SMX_OK;
  const size_t shm = 2 * (size_t)Lmax * sizeof(float);
  float* alpha = reinterpret_cast<float*>(workspace);
#define CTC_ALPHA(TT, KS_) hipLaunchKernelGGL((ctc_alpha_kernel<TT, KS_>), dim3(B), dim3(256), shm, STREAM, (const TT*)log_probs, ldlp, T, V, targets, Smax, in_len, tgt_len, blank, alpha, Lmax, nll)
#define CTC_ALPHA_T(TT) do { if (Lmax <= 256) CTC_ALPHA(TT, 1); else if (Lmax <= 512) CTC_ALPHA(TT, 2); else if (Lmax <= 1024) CTC_ALPHA(TT, 4); else if (Lmax <= 2048) CTC_ALPHA(TT, 8); else if (Lmax <= 4096) CTC_ALPHA(TT, 16); else CTC_ALPHA(TT, 32); } while (0)
  if (dtype == SMX_BF16) CTC_ALPHA_T(bf16_t); else CTC_ALPHA_T(float);
#undef CTC_ALPHA_T
#undef CTC_ALPHA
  return check_launch("smx_ctc_loss_fwd");
}

extern "C" int smx_ctc_loss_bwd(int dtype, const void* log_probs, int64_t ldlp, const int32_t* targets, const int32_t* in_len,
                                const int32_t* tgt_len, int B, int T, int V, int Smax, int blank, const float* nll,
                                const float* gscale, void* grad, int64_t ldg, void* workspace, void* stream) {
  SMX_REQUIRE(log_probs && targets && in_len && tgt_len && nll && gscale && grad && workspace, "smx_ctc_loss_bwd: null pointer");
  SMX_REQUIRE(B >= 0 && T > 0 && V > 0 && V <= 12288 && Smax >= 0 && blank >= 0 && blank < V, "smx_ctc_loss_bwd: bad sizes");
  const int Lmax = ctc_lmax(Smax);
  SMX_REQUIRE(2 * Lmax * sizeof(float) <= 64 * 1024, "smx_ctc_loss_bwd: target length %d too long (max 4095)", Smax);
  if (B == 0) return SMX_OK;
  const size_t shm = 2 * (size_t)Lmax * sizeof(float);
  float* alpha = reinterpret_cast<float*>(workspace);
  int* chain = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + ctc_alpha_bytes(B, T, Smax));
  // (dynamic LDS next to the kernel's 64 bytes of static reduction scratch)
  SMX_REQUIRE(((size_t)V + Lmax) * sizeof(float) + 64 <= 64 * 1024, "smx_ctc_loss_bwd: V + 2 Smax + 1 = %d floats (+ 64 B) do not fit the 64 KB of LDS", V + Lmax);
  hipLaunchKernelGGL(ctc_chain_kernel, dim3(B), dim3(256), 0, STREAM, targets, Smax, tgt_len, chain);
#define CTC_BETA(TT, KS_) hipLaunchKernelGGL((ctc_beta_kernel<TT, KS_>), dim3(B), dim3(256), shm, STREAM, (const TT*)log_probs, ldlp, T, V, targets, Smax, in_len, tgt_len, blank, alpha, Lmax)
#define CTC_BETA_T(TT) do { if (Lmax <= 256) CTC_BETA(TT, 1); else if (Lmax <= 512) CTC_BETA(TT, 2); else if (Lmax <= 1024) CTC_BETA(TT, 4); else if (Lmax <= 2048) CTC_BETA(TT, 8); else if (Lmax <= 4096) CTC_BETA(TT, 16); else CTC_BETA(TT, 32); } while (0)
  if (dtype == SMX_BF16) {
    CTC_BETA_T(bf16_t);
    hipLaunchKernelGGL((ctc_grad_kernel<bf16_t>), dim3(B * T), dim3(256), ((size_t)V + Lmax) * sizeof(float), STREAM, (const bf16_t*)log_probs, ldlp, T, V, targets, Smax, in_len, tgt_len, blank, alpha, Lmax, nll, gscale, chain, (bf16_t*)grad, ldg);
  } else {
    CTC_BETA_T(float);
    hipLaunchKernelGGL((ctc_grad_kernel<float>), dim3(B * T), dim3(256), ((size_t)V + Lmax) * sizeof(float), STREAM, (const float*)log_probs, ldlp, T, V, targets, Smax, in_len, tgt_len, blank, alpha, Lmax, nll, gscale, chain, (float*)grad, ldg);
  }
#undef CTC_BETA_T
#undef CTC_BETA
  return check_launch("smx_ctc_loss_bwd");
}
