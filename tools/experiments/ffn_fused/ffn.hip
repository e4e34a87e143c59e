// ffn.hip — the Conformer feed-forward module as ONE kernel per direction (bf16, d_model = 256, gfx950).
//
//   forward :  y = res + alpha * D2( D1(act(x W1^T + b1)) W2^T + b2 )        [+ LayerNorm(y) as a second output]
//   backward:  dz1 = (g W2) * act'(z1) * D1 ;  a = D1(act(z1)) (recomputed for the wgrad) ;  dx = LNbwd(dz1 W1) + res
//
// Replaces the two nn.Linear of speechbrain.nnet.attention.PositionalwiseFeedForward as the Conformer layer calls it
// (Conformer.py:458-472,507,536) and their autograd backward.
//
// Why one kernel (profiles/r02_step_c2b_v5.txt): as two GEMMs the module moves the (N, d_ffn) hidden tensor
// four times in the forward pass (Z and A written, A read back) and the two launches ran at 0.38-0.43 of the HBM roof for
// structural reasons (a 128 x 128 workgroup lives 25 K cycles for 2 K cycles of MFMA).  Here the hidden activation never
// leaves the registers: the accumulators of the first GEMM ARE the A operand of the second.
//
// Structure - "a wave owns 32 rows", the flash-attention shape with the weights in the role of K / V:
//  * workgroup = 4 waves = 128 rows, ONE workgroup per CU (launch_bounds(256, 1): 512 registers per lane);
//  * each wave keeps its 32 x 256 input panel as MFMA fragments in registers for the whole kernel (64 VGPRs) and the
//    32 x 256 output accumulators (128 AGPRs); the hidden dimension is walked in chunks of 128 units:
//        Zc (32 x 128, 64 regs) = X W1[chunk]^T      ->  bias, Z store, act, dropout, bf16  ->  8 A fragments (32 regs)
//        Y (32 x 256) += Hc W2[:, chunk]^T
//  * the weights stream L2 -> LDS by LDS-DMA (buffer_load ... lds) in 16 KB stages of 128 rows x 64 k (a 128-byte row per
//    weight row, 16-byte chunks XOR-swizzled by (row >> 1) & 7 ON THE SOURCE ADDRESS: conflict-free ds_read_b128
//    fragments), ring of 8 stages, one barrier per stage = per 16 MFMAs of every wave; nothing else is shared between
//    the waves;
//  * fragment-row permutation: the W rows of a 32-row fragment are read in the order (q0 q2 q1 q3 | q4 q6 q5 q7) of
//    their 4-row quads.  The MFMA hands lane (l31, hi) the outputs of W rows 8g + 4hi + q, so with the permuted read the
//    lane holds 8 CONSECUTIVE hidden units per register octet: 16-byte Z stores straight from the accumulators, and the
//    octet converted to bf16 is exactly the lane's A fragment of the second GEMM (k = 8 hi .. 8 hi + 7) - no LDS round
//    trip, no cross-lane traffic between the two GEMMs.  The same permutation on the W2 / W1^T side gives 16-byte
//    output stores.
#include <utility>

#include "gemm_common.h"

namespace smx {

struct FfnParams {
  const bf16_t* X; long ldx;            // fwd: module input after its LayerNorm (N, 256) | bwd: g = alpha * D2(dy) (N, 256)
  const bf16_t* W1; long ldw1;          // (F, 256)
  const bf16_t* W2; long ldw2;          // (256, F)
  const float* b1; const float* b2;     // [F], [256] or null
  bf16_t* Z; long ldz;                  // fwd: saved pre-activation (N, F) out (or null) | bwd: in
  bf16_t* A; long lda;                  // D1(act(z)) (N, F) out (or null)
  bf16_t* DZ; long lddz;                // bwd: dz1 (N, F) out
  const void* R; long ldr;              // fwd: residual (N, 256) | bwd: residual gradient (or null)
  void* Y; long ldy;                    // fwd: y (N, 256) | bwd: dx (N, 256)
  int io_f32;                           // R and Y are fp32 (the fp32 residual stream) instead of bf16
  // fwd: LayerNorm of y appended (LY null: off).  bwd: LayerNorm backward of the module's own LayerNorm
  const float* gamma; const float* beta; bf16_t* LY; long ldly; float* stats; float eps;
  const void* LX; long ldlx; const float* lstats; float* lpartial;   // bwd: LN input (N, 256), (mean, rstd), [tiles][2][256]
  int lx_f32;
  bf16_t* DX2; long lddx2; float alpha2; const uint8_t* mask2; unsigned dthresh3; float dscale3; uint64_t seed3;   // bwd second output
  int N, F, act;
  float alpha;
  unsigned dthresh1, dthresh2; float dscale1, dscale2; uint64_t seed1, seed2; const uint64_t* epoch;
  long long* dbg;                       // debug: per-wave s_memtime stamps (smx_debug_set_timing_buffer)
};

constexpr int FFN_NST = 8;              // ring stages
constexpr int FFN_STAGE = 16384;        // bytes per stage: 128 rows x 64 k bf16
typedef __attribute__((address_space(3))) void* ffn_lds_vp;

template <int N> __device__ __forceinline__ void ffn_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ffn_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// position of fragment row i (0..31) in the weight tile: quads 1 <-> 2 swapped inside every 16 rows
__device__ __forceinline__ int ffn_perm(int i) {
  const int q = (i >> 2) & 3;
  return (i & ~12) | ((q == 1 ? 2 : (q == 2 ? 1 : q)) << 2);
}

template <int ACT>
__device__ __forceinline__ float ffn_act(float v, int) { return act_fwd_c<ACT>(v); }

// 8 consecutive fp32 from LDS byte address `off`.  The address is a laundered integer on purpose: while LDS-DMA is in flight
// hipcc puts s_waitcnt vmcnt(0) in front of every LDS read whose underlying object it can name (the DMA "may alias" it) -
// that drained the whole weight ring once per epilogue group.  An address it cannot trace gets no such wait; the side
// vectors are written once, before the first DMA piece is issued.
typedef float ffn_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const ffn_f32x4* ffn_lds_f4p;
__device__ __forceinline__ void ffn_lds8(uint32_t off, float (&v)[8]) {
  const ffn_f32x4 a = *(ffn_lds_f4p)(off), b = *(ffn_lds_f4p)(off + 16);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ uint4 ffn_pack8(const float (&v)[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void ffn_unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf16_bits_to_f32(u.x & 0xffffu); f[1] = bf16_bits_to_f32(u.x >> 16);
  f[2] = bf16_bits_to_f32(u.y & 0xffffu); f[3] = bf16_bits_to_f32(u.y >> 16);
  f[4] = bf16_bits_to_f32(u.z & 0xffffu); f[5] = bf16_bits_to_f32(u.z >> 16);
  f[6] = bf16_bits_to_f32(u.w & 0xffffu); f[7] = bf16_bits_to_f32(u.w >> 16);
}
// 16-byte store through a buffer descriptor: a row beyond the tensor carries the offset 0x80000000 and is dropped by the
// hardware range check - no exec-masked branch around the store.  aux 2 = non-temporal (saved activations are not read
// again before the backward pass).
__device__ __forceinline__ void ffn_bst16(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint4& u, bool nt) {
  u32x4_t w = {u.x, u.y, u.z, u.w};
  if (nt) __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ffn_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

// MFMA with explicit register classes.  hipcc picks ONE accumulator class per function: with the 512-register budget of a
// one-wave-per-SIMD kernel every builtin MFMA accumulates in AGPRs, and because the VALU epilogue cannot read AGPRs it
// shuttled 450-2400 v_accvgpr_* per chunk between the two files (it also never used more than 128 AGPRs).  Here the
// hidden-chunk accumulators (VALU-processed every chunk) live in VGPRs, the output accumulators and the input fragments
// (touched by nothing but MFMAs) in AGPRs.  An asm statement is opaque to the compiler's hazard recogniser: the
// MFMA -> VALU read distances are covered by ffn_mfma_drain_* below, VALU-written operands by ffn_valu_settle.
__device__ __forceinline__ void ffn_mfma_zv0(f32x16& acc, const bf16x8& w, const bf16x8& x) {     // acc (VGPR) = w . x
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "a"(x));
}
__device__ __forceinline__ void ffn_mfma_zv(f32x16& acc, const bf16x8& w, const bf16x8& x) {      // acc (VGPR) += w . x (x in AGPRs)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(x));
}
__device__ __forceinline__ void ffn_mfma_ya(f32x16& acc, const bf16x8& w, const bf16x8& h) {      // acc (AGPR) += w . h
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(h));
}
// The LAST MFMA into an accumulator carries its own drain: hipcc is free to put a copy of the accumulator (a coalesced
// assignment, a live-range split) directly behind the statement that defines it and knows nothing of the 12 wait states an
// MFMA result needs before a VALU read - a separate drain statement is too late (seen: the copy was placed in FRONT of it).
__device__ __forceinline__ void ffn_mfma_zv_last(f32x16& acc, const bf16x8& w, const bf16x8& x) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(acc) : "v"(w), "a"(x));
}
__device__ __forceinline__ void ffn_mfma_ya_last(f32x16& acc, const bf16x8& w, const bf16x8& h) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+a"(acc) : "v"(w), "v"(h));
}
// an 8-pass MFMA result may be read by a VALU instruction 12 wait states after the issue at the earliest
__device__ __forceinline__ void ffn_mfma_drain_v(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void ffn_mfma_drain_v2(f32x16& a, f32x16& b) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void ffn_mfma_drain_a(f32x16 (&y)[8]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+a"(y[0]), "+a"(y[1]), "+a"(y[2]), "+a"(y[3]), "+a"(y[4]), "+a"(y[5]), "+a"(y[6]), "+a"(y[7]));
}
// VALU-written MFMA operands: two wait states before the MFMA reads them
__device__ __forceinline__ void ffn_valu_settle(bf16x8 (&h)[8]) {
  asm volatile("s_nop 1" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]));
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256, 1) void ffn_fwd_kernel(const FfnParams p) {
  constexpr int NST = FFN_NST, STAGE = FFN_STAGE, D = 256;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* b1s = reinterpret_cast<float*>(smem + NST * STAGE);
  float* b2s = b1s + p.F;
  float* gms = b2s + D;
  float* bts = gms + D;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int n = blockIdx.x * 128 + wave * 32 + l31;
  const bool rok = n < p.N;
  const long nc = rok ? n : p.N - 1;
  const int NC = p.F >> 7, S = NC * 8;
  uint32_t side0 = (uint32_t)(uintptr_t)smem + NST * STAGE;     // LDS byte address of b1 | b2 | gamma | beta
  asm volatile("" : "+v"(side0));
  const uint32_t b1o = side0, b2o = side0 + 4 * p.F, gmo = b2o + 4 * D, bto = gmo + 4 * D;
  long long* dbgp = p.dbg ? p.dbg + ((long)blockIdx.x * 4 + wave) * 16 : nullptr;
#define FFN_STAMP(k) do { if (dbgp && lane == 0) dbgp[k] = clock64(); } while (0)
  FFN_STAMP(0);

  // ---- epilogue side vectors -> LDS (visible after the first barrier of the main loop) ----
  for (int i = t; i < p.F; i += 256) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  b2s[t] = p.b2 ? p.b2[t] : 0.f;
  if (p.LY) { gms[t] = p.gamma[t]; bts[t] = p.beta[t]; }

  // ---- weight stream.  The MFMA work of a panel is 2 NC "groups" of 4 stages (16 KB each) = 16 slots of 4 MFMAs:
  //   G1(c): Z(c) = X W1[chunk c]^T, j-major: stages (half, kb2) = 64 units x 128 k (256-byte rows, chunk ^= row & 15);
  //          a slot is two k steps of the two fragments of its half
  //   G2(c): Y += H(c) W2[:, chunk c]^T: stages (kh, dh) = 128 output rows x 64 units (128-byte rows, chunk ^= (row >> 1) & 7)
  // in the order  G1(0) | G1(1) | G2(0) G1(2) | G2(1) G1(3) | ... | G2(NC-2) | G2(NC-1):  the epilogue of chunk c (bias, Z
  // store, activation, dropout, bf16) is spread over the 32 slots of [G2(c-1) G1(c+1)], so the VALU work of one chunk runs
  // in the shadow of the MFMAs of its neighbours (one wave per SIMD: nothing else could hide it).
  // A wave issues pieces w, w + 4, w + 8, w + 12 of every stage; the swizzle is applied to the SOURCE chunk. ----
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W1), (short)0,
                                                                      (int)((((long)p.F - 1) * p.ldw1 + D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W2), (short)0,
                                                                      (int)((((long)D - 1) * p.ldw2 + p.F) * 2), 0x00020000);
  // (no arrays here: a select between two captured arrays ends up as a dynamically indexed stack object, and a scratch
  // load is a vector-memory operation whose wait drains the whole LDS-DMA queue)
  const int rowa = 4 * wave + (lane >> 4);                 // G1 piece: 4 rows x 256 B; piece w + 4 j holds rows 16 j + rowa
  const int rowb = 8 * wave + (lane >> 3);                 // G2 piece: 8 rows x 128 B; piece w + 4 j holds rows 32 j + rowb
  const uint32_t va0 = (uint32_t)(((long)rowa * p.ldw1 + 8 * ((lane & 15) ^ (rowa & 15))) * 2);
  const uint32_t vb0 = (uint32_t)(((long)rowb * p.ldw2 + 8 * ((lane & 7) ^ ((rowb >> 1) & 7))) * 2);
  const uint32_t ldw1b = (uint32_t)(p.ldw1 * 2), ldw2b = (uint32_t)(p.ldw2 * 2);
  auto issue = [&](int s) {                              // (s uniform, 0 <= s < S)
    const int h = s >> 2, i4 = s & 3;
    char* dst = smem + (s & (NST - 1)) * STAGE + wave * 1024;
    bool g1;
    int c;
    if (h == 0) { g1 = true; c = 0; }
    else if (h == 2 * NC - 1) { g1 = false; c = NC - 1; }
    else if (h & 1) { g1 = true; c = (h + 1) >> 1; }
    else { g1 = false; c = (h >> 1) - 1; }
    if (g1) {
      const uint32_t so = (uint32_t)(128 * c + 64 * (i4 >> 1)) * ldw1b + (uint32_t)(256 * (i4 & 1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (ffn_lds_vp)(dst), 16, va0, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (ffn_lds_vp)(dst + 4096), 16, va0, so + 16 * ldw1b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (ffn_lds_vp)(dst + 8192), 16, va0, so + 32 * ldw1b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (ffn_lds_vp)(dst + 12288), 16, va0, so + 48 * ldw1b, 0, 0);
    } else {
      const uint32_t so = (uint32_t)(128 * (i4 & 1)) * ldw2b + (uint32_t)(256 * c + 128 * (i4 >> 1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (ffn_lds_vp)(dst), 16, vb0, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (ffn_lds_vp)(dst + 4096), 16, vb0, so + 32 * ldw2b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (ffn_lds_vp)(dst + 8192), 16, vb0, so + 64 * ldw2b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (ffn_lds_vp)(dst + 12288), 16, vb0, so + 96 * ldw2b, 0, 0);
    }
  };
  // prologue: stages 0 .. 5 (the first SYNC adds stage 6)
#pragma unroll
  for (int i = 0; i < NST - 2; ++i) issue(i);

  // ---- this wave's input panel as fragments: xf[kk] = X[n][16 kk + 8 hi .. + 7] ----
  bf16x8 xf[16];
  {
    const bf16_t* xr = p.X + nc * p.ldx + 8 * hi;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      xf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xr + 16 * kk));
      asm volatile("" : "+a"(xf[kk]));                   // (lives in the AGPRs from here on)
    }
  }
  // fragment addresses inside a stage: fragment row perm(l31) (+ 32 per fragment)
  const int pl = ffn_perm(l31);
  const uint32_t pre0 = frag_pre(pl, hi);                                  // G2 image (128-byte rows)
  const uint32_t preA = (uint32_t)(pl * 256 + ((hi ^ (pl & 15)) << 4));    // G1 image (256-byte rows)
  auto ld_g2 = [&](bf16x8 (&w)[4], const char* st, int kk) {               // 4 fragments (32 output rows each), k step kk
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = frag_kc(st, pre0 + j * 4096, kk);
  };
  auto ld_g1 = [&](bf16x8 (&w)[4], const char* st, int t) {                // 2 fragments x k steps 2 t, 2 t + 1 (of 8)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        w[2 * e + jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + jj * 8192 + (preA ^ (uint32_t)((2 * t + e) << 5))));
  };

  f32x16 Y[8];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) { Y[j] = zero16; asm volatile("" : "+a"(Y[j])); }

  const uint32_t th1 = p.dthresh1;
  const uint64_t seed1 = th1 ? epoch_seed(p.seed1, p.epoch) : 0;
  // output rows through buffer descriptors (rows >= N: offset out of range, the store is dropped)
  const __amdgpu_buffer_rsrc_t rZ = ffn_rsrc(p.Z, p.Z ? (long)p.N * p.ldz * 2 : 0), rA = ffn_rsrc(p.A, p.A ? (long)p.N * p.lda * 2 : 0);
  const uint32_t oz = rok ? (uint32_t)((long)n * p.ldz * 2) : 0x80000000u, oa = rok ? (uint32_t)((long)n * p.lda * 2) : 0x80000000u;

  // SYNC(x) = { wait for this wave's pieces of stage x; barrier; issue stage x + 6 into the slot of stage x - 2 } sits in
  // front of the LAST slot of stage x - 1: behind it every wave's pieces of stage x have landed and every wave is done with
  // stage x - 2, so the first fragments of stage x are prefetched under the last MFMAs of stage x - 1 - the fragment double
  // buffer wq rolls straight across the stage boundaries and the barrier never separates an LDS read from the MFMAs that
  // hide it.  (vmcnt retires in order: allowing only the pieces of the younger stages is conservative whatever stores sit
  // in between.)
  long long t_vm = 0, t_bar = 0;                         // (debug stamps only)
  auto sync = [&](int x) {
    const int rem = S - 1 - x;
    long long t0 = 0, t1 = 0;
    if (dbgp) t0 = clock64();
    if (rem >= 5) ffn_vm<20>(); else if (rem == 4) ffn_vm<16>(); else if (rem == 3) ffn_vm<12>();
    else if (rem == 2) ffn_vm<8>(); else if (rem == 1) ffn_vm<4>(); else ffn_vm<0>();
    if (dbgp) t1 = clock64();
    ffn_barrier();
    if (dbgp) { const long long t2 = clock64(); t_vm += t1 - t0; t_bar += t2 - t1; }
    if (x + 6 < S) issue(x + 6);
  };
  bf16x8 wq[2][4];
  sync(0);
  ld_g1(wq[0], smem, 0);

  f32x16 Z[4], Z3n;
  bf16x8 hf[8];
  // One slot = 4 MFMAs.  `gap(ActTag<m>)` is called behind MFMA m: the VALU work placed there runs while the matrix pipe
  // executes that MFMA (a wave cannot issue anything while it waits to issue its next MFMA into the busy pipe, so VALU work
  // behind a block of four MFMAs overlaps only the last of them).  hasm = false: the group does not exist in this pass (the
  // first chunk has no G2(c - 1), the last no G1(c + 1)): the weight stream is not advanced, and the MFMAs run on whatever
  // fragments sit in wq - G2 with hf == 0 adds exact zeros to Y, G1 writes accumulators nobody reads any more.  (Peeling
  // those two passes instead tripled the body and hipcc's allocation fell apart: 256 + 256 registers and 560 B of scratch.)
  // G1 group (first stage = stream stage s0); nextg1: the group behind it is a G1 group.
  auto g1_slot = [&](auto sl_tag, bool hasm, int s0, bool nextg1, auto&& gap) {
    constexpr int SL = decltype(sl_tag)::value, i4 = SL >> 2, t = SL & 3, half = i4 >> 1, kb2 = i4 & 1, cur = SL & 1, nxt = cur ^ 1;
    {
      if (hasm) {
        const char* st = smem + ((s0 + i4) & (NST - 1)) * STAGE;
        const char* stn = smem + ((s0 + i4 + 1) & (NST - 1)) * STAGE;
        if (t == 3 && s0 + i4 + 1 < S) sync(s0 + i4 + 1);
        if (t < 3) ld_g1(wq[nxt], st, t + 1);
        else if (i4 < 3) ld_g1(wq[nxt], stn, 0);
        else if (s0 + 4 < S) { if (nextg1) ld_g1(wq[nxt], stn, 0); else ld_g2(wq[nxt], stn, 0); }
      }
      const bf16x8& x0 = xf[8 * kb2 + 2 * t];
      const bf16x8& x1 = xf[8 * kb2 + 2 * t + 1];
      constexpr bool FIRST = kb2 == 0 && t == 0;
      if constexpr (half == 0) { if constexpr (FIRST) ffn_mfma_zv0(Z[0], wq[cur][0], x0); else ffn_mfma_zv(Z[0], wq[cur][0], x0); }
      else { if constexpr (FIRST) ffn_mfma_zv0(Z[2], wq[cur][0], x0); else ffn_mfma_zv(Z[2], wq[cur][0], x0); }
      gap(ActTag<0>{}); __builtin_amdgcn_sched_barrier(0);
      if constexpr (half == 0) { if constexpr (FIRST) ffn_mfma_zv0(Z[1], wq[cur][1], x0); else ffn_mfma_zv(Z[1], wq[cur][1], x0); }
      else { if constexpr (FIRST) ffn_mfma_zv0(Z3n, wq[cur][1], x0); else ffn_mfma_zv(Z3n, wq[cur][1], x0); }
      gap(ActTag<1>{}); __builtin_amdgcn_sched_barrier(0);
      constexpr bool LAST = kb2 == 1 && t == 3;              // the last two MFMAs of a half finish its two accumulators
      if constexpr (LAST) { if constexpr (half == 0) ffn_mfma_zv_last(Z[0], wq[cur][2], x1); else ffn_mfma_zv_last(Z[2], wq[cur][2], x1); }
      else { if constexpr (half == 0) ffn_mfma_zv(Z[0], wq[cur][2], x1); else ffn_mfma_zv(Z[2], wq[cur][2], x1); }
      gap(ActTag<2>{}); __builtin_amdgcn_sched_barrier(0);
      if constexpr (LAST) { if constexpr (half == 0) ffn_mfma_zv_last(Z[1], wq[cur][3], x1); else ffn_mfma_zv_last(Z3n, wq[cur][3], x1); }
      else { if constexpr (half == 0) ffn_mfma_zv(Z[1], wq[cur][3], x1); else ffn_mfma_zv(Z3n, wq[cur][3], x1); }
      gap(ActTag<3>{}); __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto g2_slot = [&](auto sl_tag, bool hasm, int s0, bool nextg1, auto&& gap, auto tail_tag) {
    constexpr bool YLAST = decltype(tail_tag)::value != 0 && ((decltype(sl_tag)::value & 3) == 3) && (decltype(sl_tag)::value >= 8);
    constexpr int SL = decltype(sl_tag)::value, i4 = SL >> 2, kk = SL & 3, kh = i4 >> 1, dh = i4 & 1, cur = SL & 1, nxt = cur ^ 1;
    {
      if (hasm) {
        const char* st = smem + ((s0 + i4) & (NST - 1)) * STAGE;
        const char* stn = smem + ((s0 + i4 + 1) & (NST - 1)) * STAGE;
        if (kk == 3 && s0 + i4 + 1 < S) sync(s0 + i4 + 1);
        if (kk < 3) ld_g2(wq[nxt], st, kk + 1);
        else if (i4 < 3) ld_g2(wq[nxt], stn, 0);
        else if (s0 + 4 < S) { if (nextg1) ld_g1(wq[nxt], stn, 0); else ld_g2(wq[nxt], stn, 0); }
      }
      if constexpr (YLAST) ffn_mfma_ya_last(Y[4 * dh + 0], wq[cur][0], hf[4 * kh + kk]); else ffn_mfma_ya(Y[4 * dh + 0], wq[cur][0], hf[4 * kh + kk]);
      gap(ActTag<0>{}); __builtin_amdgcn_sched_barrier(0);
      if constexpr (YLAST) ffn_mfma_ya_last(Y[4 * dh + 1], wq[cur][1], hf[4 * kh + kk]); else ffn_mfma_ya(Y[4 * dh + 1], wq[cur][1], hf[4 * kh + kk]);
      gap(ActTag<1>{}); __builtin_amdgcn_sched_barrier(0);
      if constexpr (YLAST) ffn_mfma_ya_last(Y[4 * dh + 2], wq[cur][2], hf[4 * kh + kk]); else ffn_mfma_ya(Y[4 * dh + 2], wq[cur][2], hf[4 * kh + kk]);
      gap(ActTag<2>{}); __builtin_amdgcn_sched_barrier(0);
      if constexpr (YLAST) ffn_mfma_ya_last(Y[4 * dh + 3], wq[cur][3], hf[4 * kh + kk]); else ffn_mfma_ya(Y[4 * dh + 3], wq[cur][3], hf[4 * kh + kk]);
      gap(ActTag<3>{}); __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- the epilogue of one register octet (8 consecutive hidden units of row n; octet o = (fragment j, half gg)) in 16
  // stages, one per MFMA gap.  Every stage is a handful of INDEPENDENT instructions (a dependent chain issues one
  // instruction per ~8 cycles on a lone wave, independent ones one per ~4), the transcendentals sit in stages of their own. ----
  float ev[8], et[8], eb[8];
  uint32_t ex[4];
  uint4 hu0 = make_uint4(0, 0, 0, 0);
  const uint32_t t16 = th1 >> 16;
  auto epi_stage = [&](auto gg_tag, int c) {
    constexpr int GG = decltype(gg_tag)::value, o = GG >> 4, G = GG & 15, j = o >> 1, gg = o & 1;
    const int u = 128 * c + 8 * hi + 32 * j + 16 * gg;
    if constexpr (G == 0) {
      ffn_lds8(b1o + 4 * u, eb);
    } else if constexpr (G == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ev[q] = Z[j][8 * gg + q] + eb[q];
      if (p.Z) ffn_bst16(rZ, oz + 2 * u, ffn_pack8(ev), true);
    } else if constexpr (ACT == SMX_ACT_SWISH && G == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) et[q] = ev[q] * -1.4426950408889634f;
    } else if constexpr (ACT == SMX_ACT_SWISH && G == 3) {
#pragma unroll
      for (int q = 0; q < 8; ++q) et[q] = __builtin_amdgcn_exp2f(et[q]);
    } else if constexpr (ACT == SMX_ACT_SWISH && G == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) et[q] += 1.0f;
    } else if constexpr (ACT == SMX_ACT_SWISH && G == 5) {
#pragma unroll
      for (int q = 0; q < 8; ++q) et[q] = __builtin_amdgcn_rcpf(et[q]);
    } else if constexpr (ACT == SMX_ACT_SWISH && G == 6) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ev[q] *= et[q];
    } else if constexpr (ACT != SMX_ACT_SWISH && G == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ev[q] = ffn_act<ACT>(ev[q], p.act);
    } else if constexpr (ACT != SMX_ACT_SWISH && G == 4) {
#pragma unroll
      for (int q = 4; q < 8; ++q) ev[q] = ffn_act<ACT>(ev[q], p.act);
    } else if constexpr (G == 7) {
      // dropout: one 32-bit hash per (even, odd) pair - dropout_apply<8> of smx_common.h spread over four stages
      if (th1) {
        const uint64_t pair0 = ((uint64_t)n * (uint64_t)p.F + (uint64_t)u) >> 1;
        const uint32_t hm = mix32((uint32_t)(pair0 >> 32) + (uint32_t)seed1) ^ (uint32_t)(seed1 >> 32);
        const uint32_t p0 = (uint32_t)pair0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ex[k] = (p0 + (uint32_t)k) ^ hm; ex[k] ^= ex[k] >> 16; }
      }
    } else if constexpr (G == 8) {
      if (th1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ex[k] = __umul24(ex[k], 0xeb352du); ex[k] ^= ex[k] >> 12; }
      }
    } else if constexpr (G == 9) {
      if (th1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ex[k] = __umul24(ex[k], 0xd2b74du); ex[k] ^= ex[k] >> 16; }
      }
    } else if constexpr (G == 10) {
      if (th1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ev[q] *= p.dscale1;
      }
    } else if constexpr (G == 11) {
      if (th1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ev[2 * k] = (ex[k] & 0xffffu) >= t16 ? ev[2 * k] : 0.f;
          ev[2 * k + 1] = (ex[k] >> 16) >= t16 ? ev[2 * k + 1] : 0.f;
        }
      }
    } else if constexpr (G == 12) {
      const uint4 hu = ffn_pack8(ev);
      if (p.A) ffn_bst16(rA, oa + 2 * u, hu, true);
      if constexpr (o == 0) hu0 = hu;                     // (G2 reads the OLD hf[0] until slot 4: committed at slot 5)
      else hf[o] = __builtin_bit_cast(bf16x8, hu);
    }
  };

  FFN_STAMP(1);
  // ---- G1(0) ----
  auto nogap = [](auto) {};
#pragma unroll
  for (int k = 0; k < 8; ++k) hf[k] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
  [&]<int... SL>(std::integer_sequence<int, SL...>) {
    (g1_slot(ActTag<SL>{}, true, 0, NC > 1, nogap), ...);
  }(std::make_integer_sequence<int, 16>{});

  // pass c: the epilogue of chunk c in the gaps of G2(c - 1) (slots 0-15) and G1(c + 1) (slots 16-31)
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    const bool has2 = c >= 1, has1 = c + 1 < NC;
    if (c == 1) FFN_STAMP(2);
    [&]<int... SL>(std::integer_sequence<int, SL...>) {
      (([&] {
         if constexpr (SL == 12) Z[3] = Z3n;               // (Z3n was drained behind its last MFMA; Z[3] was consumed in the previous pass)
         if constexpr (SL == 5) hf[0] = __builtin_bit_cast(bf16x8, hu0);
         auto gap = [&](auto m_tag) { epi_stage(ActTag<4 * SL + decltype(m_tag)::value>{}, c); };
         if constexpr (SL < 16) g2_slot(ActTag<SL>{}, has2, 8 * c, has1, gap, ActTag<0>{});
         else g1_slot(ActTag<SL - 16>{}, has1, 8 * c + 4, false, gap);
         if constexpr (SL == 15) { if (c == 1) FFN_STAMP(3); }
       }()), ...);
    }(std::make_integer_sequence<int, 32>{});
    if (c == 1) FFN_STAMP(5);
  }
  // ---- G2(NC - 1) ----
  [&]<int... SL>(std::integer_sequence<int, SL...>) {
    (g2_slot(ActTag<SL>{}, true, S - 4, false, nogap, ActTag<1>{}), ...);
  }(std::make_integer_sequence<int, 16>{});

  // ---- output: y = res + alpha * D2(Y + b2); lane (l31, hi) holds d = 32 j + 16 gg + 8 hi + q of row n ----
  ffn_mfma_drain_a(Y);
  FFN_STAMP(6);
  const uint32_t th2 = p.dthresh2;
  const uint64_t seed2 = th2 ? epoch_seed(p.seed2, p.epoch) : 0;
  const int osz = p.io_f32 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t rY = ffn_rsrc(p.Y, (long)p.N * p.ldy * osz), rL = ffn_rsrc(p.LY, p.LY ? (long)p.N * p.ldly * 2 : 0);
  const uint32_t oy = rok ? (uint32_t)((long)n * p.ldy * osz) : 0x80000000u, ol = rok ? (uint32_t)((long)n * p.ldly * 2) : 0x80000000u;
  // the residual of all the lane's outputs is requested before the first store (a load behind a store is not hoisted
  // above it: the sixteen loads came out as sixteen serial memory round trips)
  uint4 rr[16];
  if (!p.io_f32) {
    const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.R) + nc * p.ldr + 8 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) rr[i] = *reinterpret_cast<const uint4*>(rp + 16 * i);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) rr[i] = make_uint4(0, 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int d0 = 32 * j + 16 * gg + 8 * hi;
      float v[8], b[8], r[8];
      ffn_lds8(b2o + 4 * d0, b);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = Y[j][8 * gg + q] + b[q];
      if (th2) dropout_apply<8>(v, seed2, (uint64_t)n * D + (uint64_t)d0, th2, p.dscale2);
      if (p.io_f32) {
        const float* rp = reinterpret_cast<const float*>(p.R) + nc * p.ldr + d0;
        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1_ = *reinterpret_cast<const float4*>(rp + 4);
        r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1_.x; r[5] = r1_.y; r[6] = r1_.z; r[7] = r1_.w;
      } else {
        ffn_unpack8(rr[2 * j + gg], r);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[q] = r[q] + p.alpha * v[q]; sum += v[q]; Y[j][8 * gg + q] = v[q]; }
      if (p.io_f32) {
        ffn_bst16(rY, oy + 4 * d0, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])), false);
        ffn_bst16(rY, oy + 4 * d0 + 16, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])), false);
      } else {
        ffn_bst16(rY, oy + 2 * d0, ffn_pack8(v), false);
      }
    }
  if (p.LY) {
    // LayerNorm of the finished row: the two lanes (hi = 0 / 1) of a row hold 128 values each
    const float mean = (sum + __shfl_xor(sum, 32, 64)) * (1.f / 256.f);
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float dlt = Y[j][e] - mean; qq += dlt * dlt; }
    const float rstd = rsqrtf((qq + __shfl_xor(qq, 32, 64)) * (1.f / 256.f) + p.eps);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int d0 = 32 * j + 16 * gg + 8 * hi;
        float g[8], b[8], v[8];
        ffn_lds8(gmo + 4 * d0, g);
        ffn_lds8(bto + 4 * d0, b);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (Y[j][8 * gg + q] - mean) * rstd * g[q] + b[q];
        ffn_bst16(rL, ol + 2 * d0, ffn_pack8(v), false);
      }
    if (p.stats && rok && hi == 0) *reinterpret_cast<float2*>(p.stats + 2 * (long)n) = make_float2(mean, rstd);
  }
  FFN_STAMP(7);
  if (dbgp && lane == 0) { dbgp[8] = t_vm; dbgp[9] = t_bar; }
#undef FFN_STAMP
}

}  // namespace smx

using namespace smx;
extern long long* g_dbg_stamps;

extern "C" int smx_ffn_fused_ok(int dtype, int N, int D, int F) {
  return dtype == SMX_BF16 && D == 256 && F >= 128 && F % 128 == 0 && F <= 4096 && N >= 1;
}

static int ffn_fill(FfnParams& p, const smx_ffn_args* a, int N, int D, int F) {
  SMX_REQUIRE(a && a->x && a->w1 && a->w2, "smx_ffn: null pointer");
  SMX_REQUIRE(smx_ffn_fused_ok(SMX_BF16, N, D, F), "smx_ffn: needs bf16, d_model == 256, d_ffn %% 128 == 0 (got D=%d F=%d)", D, F);
  SMX_REQUIRE(aligned16(a->x) && aligned16(a->w1) && aligned16(a->w2) && a->ldx % 8 == 0 && a->ldw1 % 8 == 0 && a->ldw2 % 8 == 0,
              "smx_ffn: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
  SMX_REQUIRE((!a->z || (aligned16(a->z) && a->ldz % 8 == 0)) && (!a->a || (aligned16(a->a) && a->lda % 8 == 0)),
              "smx_ffn: z / a must be 16-byte aligned with leading dimensions %% 8 == 0");
  SMX_REQUIRE(a->drop_p1 >= 0.f && a->drop_p1 < 1.f && a->drop_p2 >= 0.f && a->drop_p2 < 1.f, "smx_ffn: 0 <= drop_p < 1");
  SMX_REQUIRE(((long)F - 1) * a->ldw1 + D < (1L << 30) && ((long)D - 1) * a->ldw2 + F < (1L << 30), "smx_ffn: weight span too large");
  SMX_REQUIRE((long)N * (a->ldz > a->lda ? a->ldz : a->lda) < (1L << 30), "smx_ffn: (N, d_ffn) tensors beyond 2 GB are not supported by the fused kernel");
  memset(&p, 0, sizeof(p));
  p.X = reinterpret_cast<const bf16_t*>(a->x); p.ldx = a->ldx;
  p.W1 = reinterpret_cast<const bf16_t*>(a->w1); p.ldw1 = a->ldw1;
  p.W2 = reinterpret_cast<const bf16_t*>(a->w2); p.ldw2 = a->ldw2;
  p.b1 = a->b1; p.b2 = a->b2;
  p.Z = reinterpret_cast<bf16_t*>(a->z); p.ldz = a->ldz;
  p.A = reinterpret_cast<bf16_t*>(a->a); p.lda = a->lda;
  p.N = N; p.F = F; p.act = a->act; p.alpha = a->alpha;
  p.dthresh1 = (unsigned)((double)a->drop_p1 * 4294967296.0); p.dscale1 = 1.f / (1.f - a->drop_p1); p.seed1 = a->drop_seed1;
  p.dthresh2 = (unsigned)((double)a->drop_p2 * 4294967296.0); p.dscale2 = 1.f / (1.f - a->drop_p2); p.seed2 = a->drop_seed2;
  p.epoch = g_step_counter;
  p.dbg = g_dbg_stamps;
  return SMX_OK;
}

template <typename K>
static int ffn_launch(K kern, const FfnParams& p, size_t lds, hipStream_t s, bool& attr_done) {
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(SMX_ELAUNCH, "smx_ffn: cannot reserve %zu bytes of LDS", lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((p.N + 127) / 128), dim3(256), lds, s, p);
  return check_launch("smx_ffn");
}

extern "C" int smx_ffn_fwd(int dtype, const smx_ffn_args* a, int N, int D, int F, void* stream) {
  SMX_REQUIRE(dtype == SMX_BF16, "smx_ffn_fwd: bf16 only");
  FfnParams p;
  int rc = ffn_fill(p, a, N, D, F);
  if (rc != SMX_OK) return rc;
  SMX_REQUIRE(a->res && a->y, "smx_ffn_fwd: res and y are required");
  const int rsz = a->io_f32 ? 4 : 2;
  SMX_REQUIRE(aligned16(a->res) && aligned16(a->y) && (a->ldr * rsz) % 16 == 0 && (a->ldy * rsz) % 16 == 0,
              "smx_ffn_fwd: res / y must be 16-byte aligned rows");
  p.R = a->res; p.ldr = a->ldr; p.Y = a->y; p.ldy = a->ldy; p.io_f32 = a->io_f32;
  if (a->lnf_y) {
    SMX_REQUIRE(a->lnf_gamma && a->lnf_beta && aligned16(a->lnf_y) && a->lnf_ldy % 8 == 0, "smx_ffn_fwd: lnf_y needs gamma / beta and aligned rows");
    p.gamma = a->lnf_gamma; p.beta = a->lnf_beta; p.LY = reinterpret_cast<bf16_t*>(a->lnf_y); p.ldly = a->lnf_ldy;
    p.stats = a->lnf_stats; p.eps = a->lnf_eps;
  }
  const size_t lds = (size_t)FFN_NST * FFN_STAGE + ((size_t)F + 3 * 256) * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static bool done[5] = {false, false, false, false, false};
  switch (a->act) {
    case SMX_ACT_SWISH: return ffn_launch(ffn_fwd_kernel<SMX_ACT_SWISH>, p, lds, s, done[2]);
    case SMX_ACT_GELU: return ffn_launch(ffn_fwd_kernel<SMX_ACT_GELU>, p, lds, s, done[1]);
    case SMX_ACT_RELU: return ffn_launch(ffn_fwd_kernel<SMX_ACT_RELU>, p, lds, s, done[4]);
    case SMX_ACT_LEAKY_RELU: return ffn_launch(ffn_fwd_kernel<SMX_ACT_LEAKY_RELU>, p, lds, s, done[3]);
    case SMX_ACT_NONE: return ffn_launch(ffn_fwd_kernel<SMX_ACT_NONE>, p, lds, s, done[0]);
  }
  return fail(SMX_EINVAL, "smx_ffn_fwd: unknown activation %d", a->act);
}
