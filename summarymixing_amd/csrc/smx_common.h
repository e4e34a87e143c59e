// smx_common.h — shared device/host helpers for the gfx950 kernels of libsmx.so.
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/smx.h"

namespace smx {

// ---------------------------------------------------------------------------------------------
// error plumbing (thread-local message; never throws)
// ---------------------------------------------------------------------------------------------
char* last_error_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
// dst[c] += sum_r partial[r][c], fixed order (rowwise.hip); shared by the act/mask backward and the GEMM colsum epilogue
void launch_colsum_partials(const float* partial, int nrows, int W, float* dst, hipStream_t stream);

#define SMX_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return ::smx::fail(SMX_EINVAL, __VA_ARGS__); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// element types
// ---------------------------------------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// fp32 -> bf16 goes through the hardware converter (v_cvt_pk_bf16_f32: round to nearest even, NaN preserving, two
// values per instruction) instead of a ~7-instruction integer sequence: the conversions were a large share of the
// VALU work of the GEMM epilogue and of every bf16 row-wise kernel.
typedef __bf16 smx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float smx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  smx_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, smx_bf16x2));
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return pack_bf16x2(f, 0.f) & 0xffffu; }
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) {
  return x;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {
  bf16_t r;
  r.v = (uint16_t)f32_to_bf16_bits(x);
  return r;
}

// 4 consecutive elements <-> 4 floats (vector width used by every row-wise kernel and the GEMM epilogue)
template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float (&f)[4]) {
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
  }
  static __device__ __forceinline__ raw pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <>
struct Vec4<bf16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float (&f)[4]) {
    f[0] = bf16_bits_to_f32(r.x & 0xffffu); f[1] = bf16_bits_to_f32(r.x >> 16);
    f[2] = bf16_bits_to_f32(r.y & 0xffffu); f[3] = bf16_bits_to_f32(r.y >> 16);
  }
  static __device__ __forceinline__ raw pack(const float (&f)[4]) {
    uint2 r;
    r.x = pack_bf16x2(f[0], f[1]);
    r.y = pack_bf16x2(f[2], f[3]);
    return r;
  }
};

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&f)[4]) {
  typename Vec4<T>::raw r = *reinterpret_cast<const typename Vec4<T>::raw*>(p);
  Vec4<T>::unpack(r, f);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&f)[4]) {
  *reinterpret_cast<typename Vec4<T>::raw*>(p) = Vec4<T>::pack(f);
}

// ---------------------------------------------------------------------------------------------
// activations (fp32 math).  gelu = exact erf GELU (torch.nn.GELU default), swish = x*sigmoid(x)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// exact-erf GELU pieces.  The device library's erff is 34 branchy VALU instructions; Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, far inside the 1e-3 parity bar) is 14 straight-line ones, and its exponential
// exp(-u^2) with u = v / sqrt(2) is the Gaussian the GELU derivative needs anyway.
// gelu_parts(v): returns erf(v / sqrt 2) and e = exp(-v^2 / 2).
__device__ __forceinline__ float gelu_parts(float v, float& e) {
  const float au = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __builtin_amdgcn_exp2f(-1.4426950408889634f * au * au);
  return copysignf(fmaf(-p * t, e, 1.0f), v);
}
__device__ __forceinline__ float gelu_fwd_(float v) { float e; return 0.5f * v * (1.0f + gelu_parts(v, e)); }
__device__ __forceinline__ float gelu_grad_(float v) {
  float e;
  const float erfv = gelu_parts(v, e);
  return 0.5f * (1.0f + erfv) + v * (0.39894228040143267794f * e);
}

__device__ __forceinline__ float act_fwd(int act, float v) {
  switch (act) {
    case SMX_ACT_GELU: return gelu_fwd_(v);
    case SMX_ACT_SWISH: return v * sigmoidf_(v);
    case SMX_ACT_LEAKY_RELU: return v >= 0.f ? v : 0.01f * v;
    case SMX_ACT_RELU: return v > 0.f ? v : 0.f;
    default: return v;
  }
}
__device__ __forceinline__ float act_grad(int act, float v) {
  switch (act) {
    case SMX_ACT_GELU: return gelu_grad_(v);
    case SMX_ACT_SWISH: {
      float s = sigmoidf_(v);
      return s * (1.0f + v * (1.0f - s));
    }
    case SMX_ACT_LEAKY_RELU: return v >= 0.f ? 1.f : 0.01f;
    case SMX_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    default: return 1.f;
  }
}

// Compile-time activation variants + a uniform dispatcher.  A run-time `switch (act)` evaluated per ELEMENT turns
// an unrolled epilogue into thousands of branches and a code size that thrashes the instruction cache (measured:
// 36 K lines of ISA, 2640 branches, 5x slower GEMM epilogue).  dispatch_act() hoists the switch: the body is
// compiled once per activation and one uniform branch selects the straight-line variant.
template <int ACT>
__device__ __forceinline__ float act_fwd_c(float v) {
  if constexpr (ACT == SMX_ACT_GELU) return gelu_fwd_(v);
  else if constexpr (ACT == SMX_ACT_SWISH) return v * sigmoidf_(v);
  else if constexpr (ACT == SMX_ACT_LEAKY_RELU) return v >= 0.f ? v : 0.01f * v;
  else if constexpr (ACT == SMX_ACT_RELU) return v > 0.f ? v : 0.f;
  else return v;
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(float v) {
  if constexpr (ACT == SMX_ACT_GELU) {
    return gelu_grad_(v);
  } else if constexpr (ACT == SMX_ACT_SWISH) {
    float s = sigmoidf_(v);
    return s * (1.0f + v * (1.0f - s));
  } else if constexpr (ACT == SMX_ACT_LEAKY_RELU) return v >= 0.f ? 1.f : 0.01f;
  else if constexpr (ACT == SMX_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  else return 1.f;
}
template <int V> struct ActTag { static constexpr int value = V; };
template <int LO, typename F, int... Is>
__device__ __forceinline__ void for_seq_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(ActTag<LO + Is>{}), ...);
}
// f(ActTag<LO>{}), ..., f(ActTag<HI - 1>{}) - a compile-time loop whose index is a constant inside f
template <int LO, int HI, typename F>
__device__ __forceinline__ void for_seq(F&& f) {
  if constexpr (LO < HI) for_seq_impl<LO>(static_cast<F&&>(f), std::make_integer_sequence<int, HI - LO>{});
}
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
  switch (act) {
    case SMX_ACT_GELU: f(ActTag<SMX_ACT_GELU>{}); break;
    case SMX_ACT_SWISH: f(ActTag<SMX_ACT_SWISH>{}); break;
    case SMX_ACT_LEAKY_RELU: f(ActTag<SMX_ACT_LEAKY_RELU>{}); break;
    case SMX_ACT_RELU: f(ActTag<SMX_ACT_RELU>{}); break;
    default: f(ActTag<SMX_ACT_NONE>{}); break;
  }
}

// counter-based dropout: keep(n, c) is a pure function of (seed, n * D + c), so the backward pass regenerates the
// forward mask from the seed instead of storing it.  Two rounds of a 32-bit multiply/xorshift mix (lowbias32) on the
// element index, XORed with a second mix of the seed for the high index bits.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
// the per-pair mixer: two rounds of (xorshift, 24-bit multiply).  v_mul_u32_u24 is a full-rate instruction, the 32-bit
// v_mul_lo_u32 of mix32 a quarter-rate one, and with two of those per pair the hash was a third of the VALU time of every
// dropout-bearing GEMM epilogue.  ONE multiply round is not enough whatever its width: idx -> C * idx is an arithmetic
// progression mod 2^32, and the keep decisions of elements 4 / 8 / 16 apart came out 0.4 correlated (tests/test_dropout_gpu.py
// ::test_dropout_mask_is_uncorrelated; this mixer: |corr| < 0.005 at every lag, like the two-round mix32).  The seed / high
// index bits go through the full mix32 (hm below).
__device__ __forceinline__ uint32_t mix32_1(uint32_t x) {
#ifdef SMX_HASH_TWO_ROUNDS
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;      // (the round-1 mixer, for A/B builds)
#else
  x ^= x >> 16; x = __umul24(x, 0xeb352du); x ^= x >> 12; x = __umul24(x, 0xd2b74du); x ^= x >> 16;
#endif
  return x;
}
// mix32_1's first multiply reads only the low 24 bits of x ^ (x >> 16): bits 24-31 of the pair index reach it only through
// bits 8-15, so pairs i and i ^ 0x02000200 (elements 2^25 + 2^10 apart - inside one 64000 x 1024 hidden tensor) hashed
// identically.  The top byte of the pair index is therefore folded into the per-group word through its own
// multiply (one v_lshrrev + v_mul_u32_u24 + v_xor per GROUP of NV elements; a group never straddles a 2^24 pair boundary).
__device__ __forceinline__ uint32_t pair_hi_mix(uint32_t pair_lo) { return __umul24(pair_lo >> 24, 0x9e3779u); }
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  // one 32-bit hash serves the two elements of an (even, odd) index pair, 16 bits each: P(drop) = (thresh>>16)/65536
  const uint64_t pair = idx >> 1;
  const uint32_t h = mix32_1((uint32_t)pair ^ mix32((uint32_t)(pair >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) ^ pair_hi_mix((uint32_t)pair));
  const uint32_t r = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return r >= (thresh >> 16);
}
// The same decisions for NV consecutive elements whose first index is a multiple of NV (NV even): the high-word mix is
// computed once per group and each 32-bit hash serves its (even, odd) pair - bit-identical to dropout_keep element by
// element, at half the (quarter-rate) integer multiplies and none of the 64-bit index arithmetic per element.
template <int NV>
__device__ __forceinline__ void dropout_apply(float (&v)[NV], uint64_t seed, uint64_t idx0, uint32_t thresh, float scale) {
  static_assert(NV % 2 == 0, "whole (even, odd) pairs");
  const uint64_t pair0 = idx0 >> 1;
  const uint32_t hm = mix32((uint32_t)(pair0 >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) ^ pair_hi_mix((uint32_t)pair0);
  const uint32_t t16 = thresh >> 16, p0 = (uint32_t)pair0;
#pragma unroll
  for (int q2 = 0; q2 < NV / 2; ++q2) {
    const uint32_t h = mix32_1((p0 + (uint32_t)q2) ^ hm);
    v[2 * q2] = (h & 0xffffu) >= t16 ? v[2 * q2] * scale : 0.f;
    v[2 * q2 + 1] = (h >> 16) >= t16 ? v[2 * q2 + 1] * scale : 0.f;
  }
}
// group form with a run-time alignment check (rows whose length is not a multiple of NV): falls back element by element
template <int NV>
__device__ __forceinline__ void dropout_apply_any(float (&v)[NV], uint64_t seed, uint64_t idx0, uint32_t thresh, float scale) {
  if ((NV % 2 == 0) && (idx0 & (uint64_t)(NV - 1)) == 0) {
    if constexpr (NV % 2 == 0) dropout_apply<NV>(v, seed, idx0, thresh, scale);
  } else {
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = dropout_keep(seed, idx0 + q, thresh) ? v[q] * scale : 0.f;
  }
}
const smx_config& cfg();                                 // the knobs of include/smx.h, read from the environment once (capi.hip)
// Optional device step counter (an explicit `epoch` argument of every call with a dropout): the seed is mixed with its
// current value, so a training step captured once in a hipGraph (constant kernel arguments) still draws fresh masks at every replay.
__device__ __forceinline__ uint64_t epoch_seed(uint64_t seed, const uint64_t* ep) {
  return ep ? seed ^ (ep[0] * 0x9E3779B97F4A7C15ull) : seed;
}
// ---- cross-lane sums on the DPP path.  __shfl_xor compiles to ds_bpermute_b32 on gfx950: every butterfly stage is a round trip
// through the LDS crossbar (~100 cycles of dependent latency, an LDS issue slot per lane pair), six of them per 64-lane sum.  The
// data-parallel-primitive modifiers of the VALU do the same inside the register file: two quad permutes, row_half_mirror and
// row_mirror leave the sum of each 16-lane row in all of its lanes; row_bcast15 / row_bcast31 carry the row sums up the wave
// (lane 31 = lanes 0-31, lane 63 = the whole wave) and v_readlane hands the total back as a wave-uniform scalar.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {     // lanes of the rows in ROW_MASK: the permuted source; the others: 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {    // every lane: the sum over its row of 16 lanes
  v += dpp_take<0xB1>(v);                                // quad_perm [1, 0, 3, 2]
  v += dpp_take<0x4E>(v);                                // quad_perm [2, 3, 0, 1]
  v += dpp_take<0x141>(v);                               // row_half_mirror
  v += dpp_take<0x140>(v);                               // row_mirror
  return v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {   // (lane: a constant) wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// sum over the 32 lanes of this lane's half wave (both halves at once), in every lane
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
  v = row16_sum(v);
  v += dpp_take<0x142, 0xa>(v);                          // row_bcast15 into rows 1 and 3: lanes 31 / 63 hold the half sums
  const float lo = lane_value(v, 31), hi = lane_value(v, 63);
  return (threadIdx.x & 32) ? hi : lo;
}
// sum over all 64 lanes, wave-uniform
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  v += dpp_take<0x142, 0xa>(v);                          // row_bcast15: rows 1, 3 += rows 0, 2
  v += dpp_take<0x143, 0xc>(v);                          // row_bcast31: rows 2, 3 += lane 31 (= rows 0 + 1)
  return lane_value(v, 63);
}
// 64-lane wave reductions (the butterfly form: every lane ends with the same bits whatever its position)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace smx
