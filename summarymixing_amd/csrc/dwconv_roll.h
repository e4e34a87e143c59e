// dwconv_roll.h — rolling register-window GLU + depthwise Conv1d (k = 31, zero padding): the Conformer conv module's
// production shape (Conformer.py:131-145,190-313).  Included by dwconv.hip.
//
// The tiled kernels of dwconv.hip build a 64-frame tile (+30 halo rows: 1.47x the bytes) in LDS and run in barrier-separated
// phases (load | window | FMA | store): 0.20-0.25 of the HBM roof.  Here ONE WAVE owns 64 channels (lane = channel) and walks a
// time segment of one utterance 16 frames at a time with its 46-47 row window in registers: no workgroup barriers, the halo
// is 30 rows per SEGMENT (128 frames: 1.23x, L2 hits next to the neighbour segment) and the rows of the next two steps are in
// flight during the current step's 2 x 496 FMAs.
//   dwconv_roll_*  (fp32): every row access is one scalar add + one buffer instruction of 4 bytes per lane.
//   dwconv_rolls_* (bf16): the vector-memory pipe retires one 64-lane instruction per 16 cycles whatever its width, so 2-byte
//     accesses cap a CU at 8 B / clk (measured: loads + stores alone 48 us for 164 MB).  Rows arrive by LDS-DMA instead
//     (buffer_load_dwordx4 ... lds: 8 rows x 128 B per instruction, no VGPRs, zero fill outside the utterance by the
//     descriptor's range check) into a per-wave ring of two steps, the window is filled by ds_read_u16, results leave through
//     a per-wave LDS tile as 16-byte stores: 10 vector-memory instructions per step instead of 80.
#pragma once

namespace smx {

constexpr int RW_STEP = 16;

template <typename T> struct RwRaw;
template <> struct RwRaw<bf16_t> { typedef unsigned short type; };
template <> struct RwRaw<float> { typedef unsigned int type; };

__device__ __forceinline__ float rw_f32(unsigned short x) { return bf16_bits_to_f32(x); }
__device__ __forceinline__ float rw_f32(unsigned int x) { return __uint_as_float(x); }

template <typename T>
__device__ __forceinline__ typename RwRaw<T>::type rw_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
  else return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
template <typename T>
__device__ __forceinline__ void rw_st(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(T) == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f32_to_bf16_bits(v), r, voff, soff, 0);
  else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// 16 rows r0 .. r0+15 of one utterance's (T, ld) matrix at lane offset voff: zeros outside [0, T).  The common case (all rows
// inside) is 16 x (s_add, buffer_load); rows are wave-uniform, so the edge case is scalar branches.
template <typename T>
__device__ __forceinline__ void rw_fetch(typename RwRaw<T>::type (&dst)[RW_STEP], __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                         int r0, int Tn, unsigned ldb) {
  if (r0 >= 0 && r0 + RW_STEP <= Tn) {
    unsigned soff = (unsigned)r0 * ldb;
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) { dst[i] = rw_ld<T>(rs, voff, soff); soff += ldb; }
  } else {
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) {
      const int r = r0 + i;
      typename RwRaw<T>::type v = 0;
      if (r >= 0 && r < Tn) v = rw_ld<T>(rs, voff, (unsigned)r * ldb);
      dst[i] = v;
    }
  }
}

// ---- Dynamic Chunk Convolution (Conformer.py:190-313; `chunk` > 0) ----------------------------------------------------
// Frame t sees no input beyond the end of its own chunk of c frames.  With m = t mod c:
//   forward and tap gradient: the right-context taps j = 16..30 (input frame t + j - 15) exist for j - 15 <= c - 1 - m,
//   input gradient at frame tau: the taps j = 16..30 (output frame tau - (j - 15)) exist for j - 15 <= tau mod c;
// the taps j = 0..15 always exist.  t mod c is wave-uniform, so the conditional taps are ONE computed scalar branch into a
// chain (no per-tap select, and the taps that do not exist are not executed): f(d) for d = 1 .. min(depth, 15).
template <int D> struct RwTap { static constexpr int value = D; };
template <int BASE, int N, typename F>
__device__ __forceinline__ void rw_taps_run(F&& f) {                    // f(BASE + 1) .. f(BASE + N)
  if constexpr (N > 0) {
    f(RwTap<BASE + 1>{});
    rw_taps_run<BASE + 1, N - 1>(f);
  }
}
// f(BASE + 1) .. f(BASE + r) for 0 <= r < 2 N (N a power of two): a binary tree of structured if / else, log2(2 N) scalar
// branches per call (a switch with fall-through costs a branch and flag moves PER TAP once hipcc has structurised it)
template <int BASE, int N, typename F>
__device__ __forceinline__ void rw_taps_tree(int r, F&& f) {
  if constexpr (N >= 1) {
    if (r >= N) {
      rw_taps_run<BASE, N>(f);
      rw_taps_tree<BASE + N, N / 2>(r - N, f);
    } else {
      rw_taps_tree<BASE, N / 2>(r, f);
    }
  }
}
template <typename F>
__device__ __forceinline__ void rw_taps_down(int depth, F&& f) {        // f(1) .. f(min(depth, 15))
  if (depth >= 15) rw_taps_run<0, 15>(f);
  else rw_taps_tree<0, 8>(depth, f);
}
// t mod c of consecutive frames, carried as a running scalar (any c >= 1): m' = m + 1, wrapped
__device__ __forceinline__ int rw_next(int m, int c) { return m + 1 == c ? 0 : m + 1; }

// (iy, bx) of this workgroup: workgroup id % 8 is its XCD; the channel tiles of one run of 4 segments sit on ONE XCD
// next to each other in dispatch order, so the 128-byte pieces of a feature row are fetched by neighbours at the same time
__device__ __forceinline__ void rw_map(int ctiles, int& iy, int& bx) {
  const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
  iy = (widx / ctiles) * 8 + xcd;
  bx = widx % ctiles;
}

template <typename T, bool CH = false>
__global__ __launch_bounds__(256) void dwconv_roll_fwd(DwParams p, int seg, int nseg, int gy) {
  constexpr int K = 31, WIN = 46, ES = (int)sizeof(T);
  typedef typename RwRaw<T>::type raw_t;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  const int item = iy * 4 + wv;
  if (iy >= gy || item >= p.B * nseg) return;
  const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
  const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
  const int ch = bx * 64 + lane;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = p.w[(long)ch * K + j];
  const float bs = p.bias ? p.bias[ch] : 0.f;
  T* Pb = const_cast<T*>(reinterpret_cast<const T*>(p.P)) + (long)b * p.T * p.ldp;
  T* Yb = reinterpret_cast<T*>(p.Y) + (long)b * p.T * p.ldy;
  const unsigned ldpb = (unsigned)p.ldp * ES, ldyb = (unsigned)p.ldy * ES;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Yb, (short)0, (int)(p.T * ldyb), 0x00020000);
  const unsigned va = (unsigned)ch * ES, vg = (unsigned)(p.D + ch) * ES;
  float win[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = 0.f;
  raw_t pa0[RW_STEP], pg0[RW_STEP], pa1[RW_STEP], pg1[RW_STEP];       // two steps of rows in flight
  const int cz = CH ? p.chunk : 1;
  int m0 = CH ? t_lo % cz : 0;                                          // (frame of the next output) mod chunk
  // window row i at step s is frame t_lo + 16 s - 15 + i; the step's 16 fetched frames land in rows 30..45
  // (a zero 'a' gives u = 0 whatever the gate: zero padding needs no flag)
  auto step = [&](raw_t (&pa)[RW_STEP], raw_t (&pg)[RW_STEP], int s) {
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) win[30 + i] = rw_f32(pa[i]) * sigmoidf_(rw_f32(pg[i]));
    if (s + 2 < nsteps) {
      rw_fetch<T>(pa, rP, va, t_lo + 15 + (s + 2) * RW_STEP, p.T, ldpb);
      rw_fetch<T>(pg, rP, vg, t_lo + 15 + (s + 2) * RW_STEP, p.T, ldpb);
    }
    if (s >= 0) {
      const int r0 = t_lo + s * RW_STEP;
      unsigned soff = (unsigned)r0 * ldyb;
#pragma unroll
      for (int o = 0; o < RW_STEP; ++o) {
        float acc = bs;
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < K; ++j) acc += w[j] * win[o + j];
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc += w[j] * win[o + j];
          rw_taps_down(cz - 1 - m0, [&](auto d) { acc += w[15 + decltype(d)::value] * win[o + 15 + decltype(d)::value]; });
          m0 = rw_next(m0, cz);
        }
        if (r0 + o < t_hi) rw_st<T>(acc, rY, va, soff);
        soff += ldyb;
      }
    }
#pragma unroll
    for (int i = 0; i < WIN - RW_STEP; ++i) win[i] = win[i + RW_STEP];
  };
  rw_fetch<T>(pa0, rP, va, t_lo + 15 - 2 * RW_STEP, p.T, ldpb);
  rw_fetch<T>(pg0, rP, vg, t_lo + 15 - 2 * RW_STEP, p.T, ldpb);
  rw_fetch<T>(pa1, rP, va, t_lo + 15 - RW_STEP, p.T, ldpb);
  rw_fetch<T>(pg1, rP, vg, t_lo + 15 - RW_STEP, p.T, ldpb);
  for (int s = -2; s < nsteps; s += 2) {
    step(pa0, pg0, s);
    if (s + 1 < nsteps) step(pa1, pg1, s + 1);
  }
}

// backward: dP = GLU'(conv^T dY), tap / bias gradient partial rows [gy][D][K + 1] (one per workgroup, the four waves folded in
// a fixed order), reduced by dw_partials_reduce_kernel or a deferred smx_reduce_jobs.
template <typename T, bool CH = false>
__global__ __launch_bounds__(256) void dwconv_roll_bwd(DwParams p, int seg, int nseg, int gy, float* __restrict__ partial) {
  constexpr int K = 31, WIN = 47, ES = (int)sizeof(T);
  typedef typename RwRaw<T>::type raw_t;
  __shared__ float wl[K][64];
  __shared__ float red[4][64 * 33];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  if (iy >= gy) return;
  const int ch = bx * 64 + lane;
  for (int j = wv; j < K; j += 4) wl[j][lane] = p.w[(long)ch * K + j];
  __syncthreads();
  const int item = iy * 4 + wv;
  float dw[K], dbs = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) dw[j] = 0.f;
  if (item < p.B * nseg) {
    const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
    const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
    T* Pb = const_cast<T*>(reinterpret_cast<const T*>(p.P)) + (long)b * p.T * p.ldp;
    T* Gb = reinterpret_cast<T*>(p.Y) + (long)b * p.T * p.ldy;                   // dY
    T* Ob = reinterpret_cast<T*>(p.dP) + (long)b * p.T * p.lddp;
    const unsigned ldpb = (unsigned)p.ldp * ES, ldyb = (unsigned)p.ldy * ES, ldob = (unsigned)p.lddp * ES;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(Gb, (short)0, (int)(p.T * ldyb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(Ob, (short)0, (int)(p.T * ldob), 0x00020000);
    const unsigned va = (unsigned)ch * ES, vg = (unsigned)(p.D + ch) * ES;
    float uw[WIN], gw[WIN], sgp[RW_STEP];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { uw[i] = 0.f; gw[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) sgp[i] = 0.f;
    raw_t pa[RW_STEP], pg[RW_STEP], pd[RW_STEP];
    const int cz = CH ? p.chunk : 1;
    int m0 = CH ? t_lo % cz : 0;                         // (frame of the step's first output) mod chunk
    // window row i at step s is frame t_lo + 16 s - 15 + i: the step's outputs (rows 15..30) are the frames fetched one step
    // earlier, the 16 frames fetched now land in rows 31..46.  The fetches of the next step are issued as soon as their
    // registers are free (dY rows right after the window took them, a / gate rows after the GLU).
    rw_fetch<T>(pd, rG, va, t_lo - RW_STEP, p.T, ldyb);
    rw_fetch<T>(pa, rP, va, t_lo - RW_STEP, p.T, ldpb);
    rw_fetch<T>(pg, rP, vg, t_lo - RW_STEP, p.T, ldpb);
    for (int s = -2; s < nsteps; ++s) {
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) gw[31 + i] = rw_f32(pd[i]);
      if (s + 1 < nsteps) rw_fetch<T>(pd, rG, va, t_lo + (s + 2) * RW_STEP, p.T, ldyb);
      if (s >= 0) {
        // du(tau) = sum_j w_j dY(tau - j + 15); GLU backward with the gate sigmoid kept from the step that fetched the frame
        float du[RW_STEP];
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) du[o] = 0.f;
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const float wj = wl[j][lane];
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) du[o] += wj * gw[30 + o - j];
          }
        } else {
          // output frames before the chunk of tau do not see it: taps j = 15 + d (output frame tau - d) for d <= tau mod c only
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float wj = wl[j][lane];
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) du[o] += wj * gw[30 + o - j];
          }
          int m = m0;
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) {
            rw_taps_down(m, [&](auto d) { du[o] += wl[15 + decltype(d)::value][lane] * gw[15 + o - decltype(d)::value]; });
            m = rw_next(m, cz);
          }
        }
        const int r0 = t_lo + s * RW_STEP;
        unsigned soff = (unsigned)r0 * ldob;
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) {
          if (r0 + o < t_hi) {
            const float sg = sgp[o];
            rw_st<T>(du[o] * sg, rO, va, soff);
            rw_st<T>(du[o] * uw[15 + o] * (1.f - sg), rO, vg, soff);            // du * a * sg * (1 - sg), u = a * sg
          }
          soff += ldob;
        }
      }
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) {
        const float sg = sigmoidf_(rw_f32(pg[i]));
        sgp[i] = sg;
        uw[31 + i] = rw_f32(pa[i]) * sg;
      }
      if (s + 1 < nsteps) {
        rw_fetch<T>(pa, rP, va, t_lo + (s + 2) * RW_STEP, p.T, ldpb);
        rw_fetch<T>(pg, rP, vg, t_lo + (s + 2) * RW_STEP, p.T, ldpb);
      }
      if (s >= 0) {
        // dw_j += sum_t dY(t) u(t + j - 15), dbias += sum_t dY(t)   (frames beyond T are zero rows)
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < K; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) sacc += gw[15 + o] * uw[o + j];
            dw[j] += sacc;
          }
        } else {
          // the forward's taps: j = 15 + d (input frame t + d) exists for d <= c - 1 - t mod c
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) sacc += gw[15 + o] * uw[o + j];
            dw[j] += sacc;
          }
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) {
            rw_taps_down(cz - 1 - m0, [&](auto d) { dw[15 + decltype(d)::value] += gw[15 + o] * uw[o + 15 + decltype(d)::value]; });
            m0 = rw_next(m0, cz);
          }
        }
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) dbs += gw[15 + o];
      }
#pragma unroll
      for (int i = 0; i < WIN - RW_STEP; ++i) { uw[i] = uw[i + RW_STEP]; gw[i] = gw[i + RW_STEP]; }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) red[wv][lane * 33 + j] = dw[j];
  red[wv][lane * 33 + K] = dbs;
  __syncthreads();
  float* out = partial + ((long)iy * p.D + bx * 64) * (K + 1);
  for (int idx = threadIdx.x; idx < 64 * (K + 1); idx += 256) {
    const int c = idx >> 5, j = idx & 31, a = c * 33 + j;
    out[idx] = ((red[0][a] + red[1][a]) + red[2][a]) + red[3][a];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16: rows through a per-wave LDS ring (LDS-DMA in, 16-byte stores out)
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* rw_lds_vp;

template <int N> __device__ __forceinline__ void rw_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector-memory operations are outstanding (n wave-uniform; rounded down to an encodable case)
__device__ __forceinline__ void rw_wait_vm(int n) {
  if (n >= 14) rw_vm<14>(); else if (n >= 10) rw_vm<10>(); else if (n >= 8) rw_vm<8>(); else if (n >= 6) rw_vm<6>();
  else if (n >= 4) rw_vm<4>(); else if (n >= 2) rw_vm<2>(); else rw_vm<0>();
}
__device__ __forceinline__ void rw_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 16 rows r0 .. r0+15 (128 B of each: this wave's 64 bf16 channels) -> 2 KB at LDS byte address dst, rows outside the buffer
// (before the utterance: the offset wraps far beyond num_records; behind it: offset >= T * ld) arrive as zeros.
// vpre = (lane / 8) * ld + (lane % 8) * 16 + this wave's column byte offset.  Two instructions, not counted by hipcc.
__device__ __forceinline__ void rw_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, unsigned vpre, int r0, unsigned ldb) {
  const unsigned s0 = (unsigned)r0 * ldb;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rw_lds_vp)dst, 16, vpre + s0, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rw_lds_vp)(dst + 1024), 16, vpre + s0 + 8 * ldb, 0, 0, 0);
}

__device__ __forceinline__ float rw_lds_bf16(const unsigned char* base, int off) {
  return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(base + off));
}

template <bool CH = false>
__global__ __launch_bounds__(256) void dwconv_rolls_fwd(DwParams p, int seg, int nseg, int gy) {
  constexpr int K = 31, WIN = 46, SLOT = 4096, OUT = 2 * SLOT, WAVE = OUT + 2048;
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][WAVE];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  const int item = iy * 4 + wv;
  if (iy >= gy || item >= p.B * nseg) return;
  const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
  const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
  const int ch = bx * 64 + lane;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = p.w[(long)ch * K + j];
  const float bs = p.bias ? p.bias[ch] : 0.f;
  bf16_t* Pb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.P)) + (long)b * p.T * p.ldp;
  bf16_t* Yb = reinterpret_cast<bf16_t*>(p.Y) + (long)b * p.T * p.ldy;
  const unsigned ldpb = (unsigned)p.ldp * 2, ldyb = (unsigned)p.ldy * 2;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Yb, (short)0, (int)(p.T * ldyb), 0x00020000);
  const unsigned vcol = (unsigned)(lane & 7) * 16 + (unsigned)bx * 128;
  const unsigned vpa = (unsigned)(lane >> 3) * ldpb + vcol, vpg = vpa + (unsigned)p.D * 2;
  const unsigned vpy = (unsigned)(lane >> 3) * ldyb + vcol;
  unsigned char* st = stage[wv];
  float win[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = 0.f;
  // window row i at step s is frame t_lo + 16 s - 15 + i; the step's 16 frames (ring slot s & 1) land in rows 30..45
  // (a zero 'a' gives u = 0 whatever the gate: zero padding needs no flag)
  auto dma = [&](int s) {
    unsigned char* d = st + (s & 1) * SLOT;
    const int r0 = t_lo + 15 + s * RW_STEP;
    rw_dma16(rP, d, vpa, r0, ldpb);
    rw_dma16(rP, d + 2048, vpg, r0, ldpb);
  };
  const int cz = CH ? p.chunk : 1;
  int m0 = CH ? t_lo % cz : 0;                          // (frame of the next output) mod chunk
  dma(-2);
  dma(-1);
  for (int s = -2; s < nsteps; ++s) {
    // vector-memory operations younger than this step's 4 DMA pieces: stores of step s-2 (2), DMA of step s+1 (4), stores of s-1 (2)
    rw_wait_vm(s < 0 ? 4 : (s >= 2 ? 2 : 0) + (s + 1 < nsteps ? 4 : 0) + (s >= 1 ? 2 : 0));
    const unsigned char* sl = st + (s & 1) * SLOT + lane * 2;
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) win[30 + i] = rw_lds_bf16(sl, i * 128) * sigmoidf_(rw_lds_bf16(sl, 2048 + i * 128));
    rw_lgkm0();                                        // the slot is read: it can take the rows of step s + 2
    if (s + 2 < nsteps) dma(s + 2);
    if (s >= 0) {
      unsigned short* ob = reinterpret_cast<unsigned short*>(st + OUT) + lane;
#pragma unroll
      for (int o = 0; o < RW_STEP; ++o) {
        float acc = bs;
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < K; ++j) acc += w[j] * win[o + j];
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc += w[j] * win[o + j];
          rw_taps_down(cz - 1 - m0, [&](auto d) { acc += w[15 + decltype(d)::value] * win[o + 15 + decltype(d)::value]; });
          m0 = rw_next(m0, cz);
        }
        ob[o * 64] = (unsigned short)f32_to_bf16_bits(acc);
      }
      asm volatile("" ::: "memory");                   // (LDS is in order within a wave: a compiler fence is enough)
      const unsigned s0 = (unsigned)(t_lo + s * RW_STEP) * ldyb;
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32v4 v = *reinterpret_cast<const u32v4*>(st + OUT + k * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rY, vpy + s0 + k * 8 * ldyb, 0, 0);   // rows >= T are dropped by the range check
      }
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < WIN - RW_STEP; ++i) win[i] = win[i + RW_STEP];
  }
}

// ABL (diagnostic, SMX_DWROLL_ABLATE): 1 = no FMAs, 2 = no DMA
template <int ABL, bool CH = false>
__global__ __launch_bounds__(256) void dwconv_rolls_bwd(DwParams p, int seg, int nseg, int gy, float* __restrict__ partial) {
  constexpr int K = 31, WIN = 47, SLOT = 6144, OUT = 2 * SLOT, WAVE = OUT + 4096;
  __shared__ float wl[K][64];
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][WAVE];            // (the 4 x 8448-byte reduction rows alias it)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  if (iy >= gy) return;
  const int ch = bx * 64 + lane;
  for (int j = wv; j < K; j += 4) wl[j][lane] = p.w[(long)ch * K + j];
  __syncthreads();
  const int item = iy * 4 + wv;
  float dw[K], dbs = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) dw[j] = 0.f;
  if (item < p.B * nseg) {
    const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
    const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
    bf16_t* Pb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.P)) + (long)b * p.T * p.ldp;
    bf16_t* Gb = reinterpret_cast<bf16_t*>(p.Y) + (long)b * p.T * p.ldy;            // dY
    bf16_t* Ob = reinterpret_cast<bf16_t*>(p.dP) + (long)b * p.T * p.lddp;
    const unsigned ldpb = (unsigned)p.ldp * 2, ldyb = (unsigned)p.ldy * 2, ldob = (unsigned)p.lddp * 2;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(Gb, (short)0, (int)(p.T * ldyb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(Ob, (short)0, (int)(p.T * ldob), 0x00020000);
    const unsigned vcol = (unsigned)(lane & 7) * 16 + (unsigned)bx * 128, vrow = (unsigned)(lane >> 3);
    const unsigned vpa = vrow * ldpb + vcol, vpg = vpa + (unsigned)p.D * 2, vpd = vrow * ldyb + vcol;
    const unsigned vo1 = vrow * ldob + vcol, vo2 = vo1 + (unsigned)p.D * 2;
    unsigned char* st = stage[wv];
    float uw[WIN], gw[WIN], sgp[RW_STEP];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { uw[i] = 0.f; gw[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) sgp[i] = 0.f;
    // window row i at step s is frame t_lo + 16 s - 15 + i: the step's outputs (rows 15..30) are the frames that arrived one
    // step earlier, the 16 frames of ring slot s & 1 land in rows 31..46
    auto dma = [&](int s) {
      if (ABL == 2) return;
      unsigned char* d = st + (s & 1) * SLOT;
      const int r0 = t_lo + (s + 1) * RW_STEP;
      rw_dma16(rG, d, vpd, r0, ldyb);
      rw_dma16(rP, d + 2048, vpa, r0, ldpb);
      rw_dma16(rP, d + 4096, vpg, r0, ldpb);
    };
    const int cz = CH ? p.chunk : 1;
    int m0 = CH ? t_lo % cz : 0;                         // (frame of the step's first output) mod chunk
    dma(-2);
    dma(-1);
    for (int s = -2; s < nsteps; ++s) {
      // younger than this step's 6 DMA pieces: stores of step s-2 (4), DMA of step s+1 (6), stores of step s-1 (4)
      rw_wait_vm(s < 0 ? 6 : (s >= 2 ? 4 : 0) + (s + 1 < nsteps ? 6 : 0) + (s >= 1 ? 4 : 0));
      const unsigned char* sl = st + (s & 1) * SLOT + lane * 2;
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) gw[31 + i] = rw_lds_bf16(sl, i * 128);
      float sgn[RW_STEP];
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) {
        sgn[i] = sigmoidf_(rw_lds_bf16(sl, 4096 + i * 128));
        uw[31 + i] = rw_lds_bf16(sl, 2048 + i * 128) * sgn[i];
      }
      rw_lgkm0();                                      // the slot is read: it can take the rows of step s + 2
      if (s + 2 < nsteps) dma(s + 2);
      if (s >= 0) {
        // du(tau) = sum_j w_j dY(tau - j + 15); GLU backward with the gate sigmoid kept from the step that brought the frame
        float du[RW_STEP];
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) du[o] = 0.f;
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < (ABL == 1 ? 1 : K); ++j) {
            const float wj = wl[j][lane];
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) du[o] += wj * gw[30 + o - j];
          }
        } else {
          // output frames before the chunk of tau do not see it: taps j = 15 + d (output frame tau - d) for d <= tau mod c only
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float wj = wl[j][lane];
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) du[o] += wj * gw[30 + o - j];
          }
          int m = m0;
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) {
            rw_taps_down(m, [&](auto d) { du[o] += wl[15 + decltype(d)::value][lane] * gw[15 + o - decltype(d)::value]; });
            m = rw_next(m, cz);
          }
        }
        unsigned short* ob = reinterpret_cast<unsigned short*>(st + OUT) + lane;
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) {
          const float sg = sgp[o];
          ob[o * 64] = (unsigned short)f32_to_bf16_bits(du[o] * sg);
          ob[1024 + o * 64] = (unsigned short)f32_to_bf16_bits(du[o] * uw[15 + o] * (1.f - sg));   // du a sg (1 - sg), u = a sg
        }
        asm volatile("" ::: "memory");                 // (LDS is in order within a wave: a compiler fence is enough)
        const unsigned s0 = (unsigned)(t_lo + s * RW_STEP) * ldob;
        typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < 2; ++k) {                  // rows >= T are dropped by the descriptor's range check
          const u32v4 v1 = *reinterpret_cast<const u32v4*>(st + OUT + k * 1024 + lane * 16);
          const u32v4 v2 = *reinterpret_cast<const u32v4*>(st + OUT + 2048 + k * 1024 + lane * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v1, rO, vo1 + s0 + k * 8 * ldob, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(v2, rO, vo2 + s0 + k * 8 * ldob, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) sgp[i] = sgn[i];
      if (s >= 0) {
        // dw_j += sum_t dY(t) u(t + j - 15), dbias += sum_t dY(t)   (frames beyond T are zero rows)
        if constexpr (!CH) {
#pragma unroll
          for (int j = 0; j < (ABL == 1 ? 1 : K); ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) sacc += gw[15 + o] * uw[o + j];
            dw[j] += sacc;
          }
        } else {
          // the forward's taps: j = 15 + d (input frame t + d) exists for d <= c - 1 - t mod c
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < RW_STEP; ++o) sacc += gw[15 + o] * uw[o + j];
            dw[j] += sacc;
          }
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) {
            rw_taps_down(cz - 1 - m0, [&](auto d) { dw[15 + decltype(d)::value] += gw[15 + o] * uw[o + 15 + decltype(d)::value]; });
            m0 = rw_next(m0, cz);
          }
        }
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) dbs += gw[15 + o];
      }
#pragma unroll
      for (int i = 0; i < WIN - RW_STEP; ++i) { uw[i] = uw[i + RW_STEP]; gw[i] = gw[i + RW_STEP]; }
    }
  }
  __syncthreads();                                     // every wave is done with its ring: the reduction rows take its place
  float* red = reinterpret_cast<float*>(&stage[0][0]);
#pragma unroll
  for (int j = 0; j < K; ++j) red[wv * 2112 + lane * 33 + j] = dw[j];
  red[wv * 2112 + lane * 33 + K] = dbs;
  __syncthreads();
  float* out = partial + ((long)iy * p.D + bx * 64) * (K + 1);
  for (int idx = threadIdx.x; idx < 64 * (K + 1); idx += 256) {
    const int a = (idx >> 5) * 33 + (idx & 31);
    out[idx] = ((red[a] + red[2112 + a]) + red[2 * 2112 + a]) + red[3 * 2112 + a];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// CSGU (Branchformer cgMLP): y = (conv_reflect(v) + bias) * gate, bf16, same per-wave rolling structure.  The reflect padding
// is a row MAP applied to the DMA source (frame -r reads frame r, frame T-1+r reads frame T-1-r), so the window simply holds
// the padded signal.  Backward: dgate = dY * (conv(v) + bias) (forward recomputed from the window), c = dY * gate,
// dv = conv^T(c) for the real frames; the gradient that the mirrored virtual frames send back to frames 1..15 and
// T-16..T-2 is added by dwconv_csgu_fold_kernel afterwards (30 rows per utterance).
// ---------------------------------------------------------------------------------------------------------------------
// 16 rows r0 .. r0+15 mirrored into [0, T) (rows still outside: zeros)
__device__ __forceinline__ void rw_dma16_reflect(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, unsigned vcol, int vrow, int r0,
                                                 int Tn, unsigned ldb) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int r = r0 + 8 * k + vrow;
    r = r < 0 ? -r : (r >= Tn ? 2 * (Tn - 1) - r : r);
    const unsigned off = (r >= 0 && r < Tn) ? (unsigned)r * ldb + vcol : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rw_lds_vp)(dst + 1024 * k), 16, off, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void dwconv_rollc_fwd(DwParams p, int seg, int nseg, int gy) {
  constexpr int K = 31, WIN = 46, SLOT = 4096, OUT = 2 * SLOT, WAVE = OUT + 2048;
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][WAVE];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  const int item = iy * 4 + wv;
  if (iy >= gy || item >= p.B * nseg) return;
  const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
  const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
  const int ch = bx * 64 + lane;
  float w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = p.w[(long)ch * K + j];
  const float bs = p.bias ? p.bias[ch] : 0.f;
  bf16_t* Pb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.P)) + (long)b * p.T * p.ldp;
  bf16_t* Gb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.gate)) + (long)b * p.T * p.ldg;
  bf16_t* Yb = reinterpret_cast<bf16_t*>(p.Y) + (long)b * p.T * p.ldy;
  const unsigned ldpb = (unsigned)p.ldp * 2, ldgb = (unsigned)p.ldg * 2, ldyb = (unsigned)p.ldy * 2;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(Gb, (short)0, (int)(p.T * ldgb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Yb, (short)0, (int)(p.T * ldyb), 0x00020000);
  const unsigned vcol = (unsigned)(lane & 7) * 16 + (unsigned)bx * 128;
  const int vrow = lane >> 3;
  const unsigned vpg = (unsigned)vrow * ldgb + vcol, vpy = (unsigned)vrow * ldyb + vcol;
  const uint32_t dthresh = p.dthresh;
  const uint64_t dseed = dthresh ? epoch_seed(p.dseed, p.epoch) : 0;
  unsigned char* st = stage[wv];
  float win[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = 0.f;
  // ring slot s & 1: [v rows t_lo + 15 + 16 s ..][gate rows t_lo + 16 s .. (the step's OUTPUT frames)]
  auto dma = [&](int s) {
    unsigned char* d = st + (s & 1) * SLOT;
    rw_dma16_reflect(rP, d, vcol, vrow, t_lo + 15 + s * RW_STEP, p.T, ldpb);
    rw_dma16(rG, d + 2048, vpg, t_lo + s * RW_STEP, ldgb);
  };
  dma(-2);
  dma(-1);
  for (int s = -2; s < nsteps; ++s) {
    rw_wait_vm(s < 0 ? 4 : (s >= 2 ? 2 : 0) + (s + 1 < nsteps ? 4 : 0) + (s >= 1 ? 2 : 0));
    const unsigned char* sl = st + (s & 1) * SLOT + lane * 2;
    float gt[RW_STEP];
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) { win[30 + i] = rw_lds_bf16(sl, i * 128); gt[i] = rw_lds_bf16(sl, 2048 + i * 128); }
    rw_lgkm0();
    if (s + 2 < nsteps) dma(s + 2);
    if (s >= 0) {
      unsigned short* ob = reinterpret_cast<unsigned short*>(st + OUT) + lane;
#pragma unroll
      for (int o = 0; o < RW_STEP; ++o) {
        float acc = bs;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += w[j] * win[o + j];
        float yv = acc * gt[o];
        if (dthresh) {                                   // the CSGU's own dropout: mask index = global row * D + channel
          const uint64_t idx = ((uint64_t)b * p.T + (uint64_t)(t_lo + s * RW_STEP + o)) * (uint64_t)p.D + ch;
          yv = dropout_keep(dseed, idx, dthresh) ? yv * p.dscale : 0.f;
        }
        ob[o * 64] = (unsigned short)f32_to_bf16_bits(yv);
      }
      asm volatile("" ::: "memory");
      const unsigned s0 = (unsigned)(t_lo + s * RW_STEP) * ldyb;
      typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32v4 v = *reinterpret_cast<const u32v4*>(st + OUT + k * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rY, vpy + s0 + k * 8 * ldyb, 0, 0);
      }
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < WIN - RW_STEP; ++i) win[i] = win[i + RW_STEP];
  }
}

__global__ __launch_bounds__(256) void dwconv_rollc_bwd(DwParams p, int seg, int nseg, int gy, float* __restrict__ partial) {
  constexpr int K = 31, WIN = 47, SLOT = 6144, OUT = 2 * SLOT, WAVE = OUT + 4096;
  __shared__ float wl[K][64];
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][WAVE];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int iy, bx;
  rw_map(p.D / 64, iy, bx);
  if (iy >= gy) return;
  const int ch = bx * 64 + lane;
  for (int j = wv; j < K; j += 4) wl[j][lane] = p.w[(long)ch * K + j];
  __syncthreads();
  const int item = iy * 4 + wv;
  float dw[K], dbs = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) dw[j] = 0.f;
  if (item < p.B * nseg) {
    const int b = item / nseg, t_lo = (item % nseg) * seg, t_hi = min(p.T, t_lo + seg);
    const int nsteps = (t_hi - t_lo + RW_STEP - 1) / RW_STEP;
    const float bs = p.bias ? p.bias[ch] : 0.f;
    bf16_t* Pb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.P)) + (long)b * p.T * p.ldp;       // v
    bf16_t* Gb = const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.gate)) + (long)b * p.T * p.ldg;
    bf16_t* Yb = reinterpret_cast<bf16_t*>(p.Y) + (long)b * p.T * p.ldy;                                    // dY
    bf16_t* Ob = reinterpret_cast<bf16_t*>(p.dP) + (long)b * p.T * p.lddp;                                  // dv
    bf16_t* Qb = reinterpret_cast<bf16_t*>(p.dgate) + (long)b * p.T * p.lddg;
    const unsigned ldpb = (unsigned)p.ldp * 2, ldgb = (unsigned)p.ldg * 2, ldyb = (unsigned)p.ldy * 2;
    const unsigned ldob = (unsigned)p.lddp * 2, ldqb = (unsigned)p.lddg * 2;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pb, (short)0, (int)(p.T * ldpb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(Gb, (short)0, (int)(p.T * ldgb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Yb, (short)0, (int)(p.T * ldyb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(Ob, (short)0, (int)(p.T * ldob), 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(Qb, (short)0, (int)(p.T * ldqb), 0x00020000);
    const unsigned vcol = (unsigned)(lane & 7) * 16 + (unsigned)bx * 128;
    const int vrow = lane >> 3;
    const unsigned vpd = (unsigned)vrow * ldyb + vcol, vpgt = (unsigned)vrow * ldgb + vcol;
    const unsigned vo = (unsigned)vrow * ldob + vcol, vq = (unsigned)vrow * ldqb + vcol;
    unsigned char* st = stage[wv];
    float vw[WIN], cw[WIN], dyp[RW_STEP];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { vw[i] = 0.f; cw[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < RW_STEP; ++i) dyp[i] = 0.f;
    // window row i at step s is frame t_lo + 16 s - 15 + i (vw: the reflect-padded v, cw: c = dY * gate, zero outside the
    // utterance); outputs = rows 15..30 = the frames that arrived one step earlier
    auto dma = [&](int s) {
      unsigned char* d = st + (s & 1) * SLOT;
      const int r0 = t_lo + (s + 1) * RW_STEP;
      rw_dma16(rY, d, vpd, r0, ldyb);
      rw_dma16(rG, d + 2048, vpgt, r0, ldgb);
      rw_dma16_reflect(rP, d + 4096, vcol, vrow, r0, p.T, ldpb);
    };
    dma(-2);
    dma(-1);
    for (int s = -2; s < nsteps; ++s) {
      rw_wait_vm(s < 0 ? 6 : (s >= 2 ? 4 : 0) + (s + 1 < nsteps ? 6 : 0) + (s >= 1 ? 4 : 0));
      const unsigned char* sl = st + (s & 1) * SLOT + lane * 2;
      float dyn[RW_STEP];
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) {
        dyn[i] = rw_lds_bf16(sl, i * 128);
        cw[31 + i] = dyn[i] * rw_lds_bf16(sl, 2048 + i * 128);
        vw[31 + i] = rw_lds_bf16(sl, 4096 + i * 128);
      }
      rw_lgkm0();
      if (s + 2 < nsteps) dma(s + 2);
      if (s >= 0) {
        float du[RW_STEP], cv[RW_STEP];
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) { du[o] = 0.f; cv[o] = bs; }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float wj = wl[j][lane];
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) { du[o] += wj * cw[30 + o - j]; cv[o] += wj * vw[o + j]; }
        }
        unsigned short* ob = reinterpret_cast<unsigned short*>(st + OUT) + lane;
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) {
          ob[o * 64] = (unsigned short)f32_to_bf16_bits(du[o]);
          ob[1024 + o * 64] = (unsigned short)f32_to_bf16_bits(dyp[o] * cv[o]);
        }
        asm volatile("" ::: "memory");
        const int r0 = t_lo + s * RW_STEP;
        typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const u32v4 v1 = *reinterpret_cast<const u32v4*>(st + OUT + k * 1024 + lane * 16);
          const u32v4 v2 = *reinterpret_cast<const u32v4*>(st + OUT + 2048 + k * 1024 + lane * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v1, rO, vo + (unsigned)(r0 + 8 * k) * ldob, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(v2, rQ, vq + (unsigned)(r0 + 8 * k) * ldqb, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < RW_STEP; ++i) dyp[i] = dyn[i];
      if (s >= 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          float sacc = 0.f;
#pragma unroll
          for (int o = 0; o < RW_STEP; ++o) sacc += cw[15 + o] * vw[o + j];
          dw[j] += sacc;
        }
#pragma unroll
        for (int o = 0; o < RW_STEP; ++o) dbs += cw[15 + o];
      }
#pragma unroll
      for (int i = 0; i < WIN - RW_STEP; ++i) { vw[i] = vw[i + RW_STEP]; cw[i] = cw[i + RW_STEP]; }
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(&stage[0][0]);
#pragma unroll
  for (int j = 0; j < K; ++j) red[wv * 2112 + lane * 33 + j] = dw[j];
  red[wv * 2112 + lane * 33 + K] = dbs;
  __syncthreads();
  float* out = partial + ((long)iy * p.D + bx * 64) * (K + 1);
  for (int idx = threadIdx.x; idx < 64 * (K + 1); idx += 256) {
    const int a = (idx >> 5) * 33 + (idx & 31);
    out[idx] = ((red[a] + red[2112 + a]) + red[2 * 2112 + a]) + red[3 * 2112 + a];
  }
}

// The gradient the mirrored virtual frames send back (T >= 16), c = dY * gate:
//   top:    dv[tau]         += sum_{j=0}^{15-tau} w_j      c[15 - tau - j]        tau = 1..15   (virtual frame -tau)
//   bottom: dv[T-1-kk]      += sum_{m=0}^{15-kk}  w_{15+kk+m} c[T-1-m]            kk  = 1..15   (virtual frame T-1+kk)
// One thread owns two adjacent channels of one utterance and does both edges one after the other (for T < 31 a frame can
// receive both terms): 30 rows of c, 30 taps, 2 x 120 FMAs, 30 read-modify-writes of 4 bytes.
__global__ __launch_bounds__(64) void dwconv_csgu_fold_kernel(DwParams p) {
  constexpr int K = 31;
  const int c2 = (blockIdx.x * 64 + threadIdx.x) * 2, b = blockIdx.y;
  if (c2 >= p.D) return;
  const bf16_t* dY = reinterpret_cast<const bf16_t*>(p.Y) + (long)b * p.T * p.ldy + c2;
  const bf16_t* G = reinterpret_cast<const bf16_t*>(p.gate) + (long)b * p.T * p.ldg + c2;
  bf16_t* dV = reinterpret_cast<bf16_t*>(p.dP) + (long)b * p.T * p.lddp + c2;
  auto ld2 = [](const bf16_t* q, float& lo, float& hi) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(q);
    lo = bf16_bits_to_f32(u & 0xffffu); hi = bf16_bits_to_f32(u >> 16);
  };
#pragma unroll 1
  for (int edge = 0; edge < 2; ++edge) {
    float c0[15], c1[15], w0[15], w1[15];
#pragma unroll
    for (int m = 0; m < 15; ++m) {
      const long t = edge == 0 ? m : p.T - 1 - m;          // c row m of this edge
      float y0, y1, g0, g1;
      ld2(dY + t * p.ldy, y0, y1);
      ld2(G + t * p.ldg, g0, g1);
      c0[m] = y0 * g0; c1[m] = y1 * g1;
      const int j = edge == 0 ? m : 16 + m;                 // taps 0..14 (top) / 16..30 (bottom)
      w0[m] = p.w[(long)c2 * K + j]; w1[m] = p.w[(long)(c2 + 1) * K + j];
    }
#pragma unroll
    for (int q = 1; q <= 15; ++q) {                        // q = tau (top) or kk (bottom)
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int m = 0; m <= 15 - q; ++m) {
        // top: tap j = m, row 15 - q - m;  bottom: tap 15 + q + m = w[..][q - 1 + m], row m
        const int wi = edge == 0 ? m : q - 1 + m, ci = edge == 0 ? 15 - q - m : m;
        a0 += w0[wi] * c0[ci]; a1 += w1[wi] * c1[ci];
      }
      bf16_t* o = dV + (long)(edge == 0 ? q : p.T - 1 - q) * p.lddp;
      float o0, o1;
      ld2(o, o0, o1);
      *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(o0 + a0, o1 + a1);
    }
  }
}

// time segment per wave: 128 frames (8 steps; 30 halo rows = 1.23x) unless that leaves the chip short of waves
inline void roll_geometry(int B, int T, int D, int* seg, int* nseg, int* gy) {
  int s = 128;
  while (s > 32 && (long)B * ((T + s - 1) / s) * (D / 64) < 2048) s >>= 1;    // < 2 waves per SIMD: shorter segments
  *seg = s;
  *nseg = (T + s - 1) / s;
  *gy = (int)(((long)B * *nseg + 3) / 4);
}

}  // namespace smx
