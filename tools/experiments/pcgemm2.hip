// pcgemm2.hip — EXPERIMENT (not part of libsmx.so): producer / consumer GEMM for the output-heavy K = 256 shapes.
//
// Question (DESIGN_APPENDIX.md §5 / §7): a 128 x 128 workgroup of gemm.hip lives 25.5 K cycles for 2 K cycles of MFMA work; its four
// waves load, multiply, stage and run the epilogue one after the other, and the three workgroups of a CU do it in phase.
// Does it pay to give the two halves to DIFFERENT waves of one persistent workgroup?
//
//   768 threads = 12 waves per CU (one workgroup per CU, 133 KB of LDS):
//     waves 0-3  (MFMA group):  64 x 64 accumulators each; operands HBM/L2 -> registers -> LDS (two LDS slots, two register
//                               stages: the stream of (tile, K step) pairs runs ahead across tile boundaries)
//     waves 4-11 (epilogue group, 512 threads): bias + Swish + two bf16 stores (Z, H) of the PREVIOUS tile, 32 staged rows
//                               per interval, out of a full-tile fp32 staging area
//   One s_barrier per interval for everybody: per tile K/64 = 4 K-step intervals (epilogue phases 0-3 of the previous tile
//   run beside them) + 1 interval in which the MFMA group dumps its accumulators into the staging area.
//
// Fixed function on purpose: NT, bf16, K % 128 == 0 (>= 256), M % 128 == 0, bias + Swish, Z and H outputs.
// tools/pcgemm_bench.py builds it (hipcc -shared), checks it against smx_gemm and times both.
#include <hip/hip_runtime.h>

#include "gemm_common.h"

namespace smx {

constexpr int PC_STG_LD = 128 * 4 + 16;                  // fp32 staging row (16 B pad)
constexpr int PC_OP = 16384;                             // one operand slot: 128 rows x 64 k x 2 B
constexpr int PC_STG = 128 * PC_STG_LD;                  // 67584
constexpr int PC_LDS = 4 * PC_OP + PC_STG + 1024;        // A0 A1 B0 B1 | staging | bias[2][128]

struct PcParams {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* Z; bf16_t* H;
  long lda, ldw, ldz, ldh;
  int N, M, K, tiles_m, ntiles, ablate;                  // ablate: 1 = no epilogue work, 2 = no MFMA, 4 = no loads
};

__global__ __launch_bounds__(768) void pc_gemm_kernel(PcParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int q0 = blockIdx.x * per, q1 = min(p.ntiles, q0 + per);
  if (q0 >= q1) return;
  const int nt = q1 - q0, nk = p.K / 64, total = nt * nk;
  char* As = smem;
  char* Bs = smem + 2 * PC_OP;
  char* stg = smem + 4 * PC_OP;
  float* side = reinterpret_cast<float*>(stg + PC_STG);

  if (wave < 4) {
    // ------------------------------------------------------------------------------------------------ MFMA group
    const int wn = wave >> 1, wm = wave & 1, l31 = lane & 31, hi = lane >> 5;
    BufStage<bf16_t, true, 128> bufa, bufb;
    uint4 ra[2][4], rb[2][4];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[s_][i] = rb[s_][i] = make_uint4(0, 0, 0, 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint32_t fpa[2], fpb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { fpa[i] = frag_pre(wn * 64 + i * 32 + l31, hi); fpb[i] = frag_pre(wm * 64 + i * 32 + l31, hi); }
    // stream position s = (tile, K step): request its operands
    auto issue = [&](int s, uint4 (&a)[4], uint4 (&b)[4]) {
      if (s >= total || (p.ablate & 4)) return;
      const int q = q0 + s / nk, ks = s % nk;
      bufa.init(p.A, p.lda, (q / p.tiles_m) * 128, p.N, p.K, t);
      bufb.init(p.W, p.ldw, (q % p.tiles_m) * 128, p.M, p.K, t);
      bufa.load(a, ks * 64);
      bufb.load(b, ks * 64);
    };
    // interval of stream position s (slot P = s & 1 is multiplied, slot P ^ 1 receives stream s + 1)
    auto kstep = [&](auto par, int s) {
      constexpr int P = decltype(par)::value, Q = P ^ 1;
      lds_barrier();
      if (s + 1 < total) {
        stage_store<bf16_t, true, 128>(ra[Q], As + Q * PC_OP, t);
        stage_store<bf16_t, true, 128>(rb[Q], Bs + Q * PC_OP, t);
        issue(s + 3, ra[Q], rb[Q]);
      }
      if (p.ablate & 2) return;
      const char* a_ = As + P * PC_OP;
      const char* b_ = Bs + P * PC_OP;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[i] = frag_kc(a_, fpa[i], kk); fb[i] = frag_kc(b_, fpb[i], kk); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
    };
    issue(0, ra[0], rb[0]);
    issue(1, ra[1], rb[1]);
    stage_store<bf16_t, true, 128>(ra[0], As, t);
    stage_store<bf16_t, true, 128>(rb[0], Bs, t);
    issue(2, ra[0], rb[0]);
    for (int tl = 0; tl <= nt; ++tl) {                   // (tl == nt: the drain tile - barriers only)
      if (tl < nt) {
        for (int ks = 0; ks < nk; ks += 2) {
          kstep(ActTag<0>{}, tl * nk + ks);
          kstep(ActTag<1>{}, tl * nk + ks + 1);
        }
      } else {
        for (int ks = 0; ks < nk; ++ks) lds_barrier();
      }
      lds_barrier();                                     // dump interval: the staging area is free (phases 0-3 are done)
      if (tl < nt) {
        const int q = q0 + tl, m0 = (q % p.tiles_m) * 128;
        if (t < 128) side[(tl & 1) * 128 + t] = p.bias ? p.bias[m0 + t] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              *reinterpret_cast<float4*>(stg + (wn * 64 + i * 32 + l31) * PC_STG_LD + (wm * 64 + j * 32 + g * 8 + hi * 4) * 4) =
                  make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) acc[i][j][g * 4 + q4] = 0.f;
            }
      }
    }
  } else {
    // -------------------------------------------------------------------------------------------- epilogue group
    const int te = t - 256, c = (te & 15) * 8, rr = te >> 4;        // 512 threads: 32 rows x 16 items of 8 columns
    for (int tl = 0; tl <= nt; ++tl) {
      for (int ks = 0; ks < nk; ++ks) {
        lds_barrier();
        if (tl == 0 || ks >= 4 || (p.ablate & 1)) continue;
        const int q = q0 + tl - 1, n0 = (q / p.tiles_m) * 128, m0 = (q % p.tiles_m) * 128;
        const int r = ks * 32 + rr, n = n0 + r;
        const float4 a0 = *reinterpret_cast<const float4*>(stg + r * PC_STG_LD + c * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(stg + r * PC_STG_LD + c * 4 + 16);
        const float* bs = side + ((tl - 1) & 1) * 128 + c;
        const float4 b0 = *reinterpret_cast<const float4*>(bs), b1 = *reinterpret_cast<const float4*>(bs + 4);
        float v[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
        if (n < p.N) {
          st_elems_nt<bf16_t, 8>(p.Z + (long)n * p.ldz + m0 + c, v);
          act_fwd_n<SMX_ACT_SWISH, 8>(v);
          st_elems<bf16_t, 8>(p.H + (long)n * p.ldh + m0 + c, v);
        }
      }
      lds_barrier();
    }
  }
}

}  // namespace smx

extern "C" int pc_gemm(const void* A, long lda, const void* W, long ldw, const float* bias, void* Z, long ldz, void* H, long ldh,
                       int N, int M, int K, int blocks, int ablate, void* stream) {
  using namespace smx;
  if (K % 128 != 0 || K < 256 || M % 128 != 0 || N < 1) return -1;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS) != hipSuccess)
      return -2;
    attr = true;
  }
  PcParams p;
  p.A = reinterpret_cast<const bf16_t*>(A); p.W = reinterpret_cast<const bf16_t*>(W); p.bias = bias;
  p.Z = reinterpret_cast<bf16_t*>(Z); p.H = reinterpret_cast<bf16_t*>(H);
  p.lda = lda; p.ldw = ldw; p.ldz = ldz; p.ldh = ldh; p.N = N; p.M = M; p.K = K;
  p.tiles_m = M / 128; p.ntiles = ((N + 127) / 128) * p.tiles_m; p.ablate = ablate;
  hipLaunchKernelGGL(pc_gemm_kernel, dim3(blocks > 0 ? blocks : 256), dim3(768), PC_LDS, reinterpret_cast<hipStream_t>(stream), p);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
