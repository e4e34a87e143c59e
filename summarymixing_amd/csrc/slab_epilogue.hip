// slab_epilogue.hip — the reducer of the split-K Linears of a small batch (round 6, with smx_gemm_panel_slabs):
//
//   v[n, :]   = sum_s slab[s][n, :]                    (float32 partial products, FIXED order s = 0, 1, ...: bit-reproducible, no atomics)
//   C[n, :]   = res[n, :] + alpha * mask[n] * D(act(v + bias))        [Z = v + bias saved for the backward pass]
//   Y[n, :]   = lact(LayerNorm(C[n, :]))  [+ Y2 = LayerNorm2(Y)]      the LayerNorm(s) that follow the Linear in the encoder layer
//
// i.e. the epilogue of smx_gemm (include/smx.h: smx_epilogue: bias, act + z, dropout, alpha, row_mask, res, SMX_EPI_LN_FWD, lnf2_*)
// applied by a ROW kernel - one wave per row, the row in registers - so that it costs what the standalone LayerNorm launch that
// would follow the GEMM costs anyway: at 3750 frames every launch of the replayed step is >= 4.5 us whatever it does, and a
// LayerNorm-fused GEMM tile (128 whole rows per workgroup) would leave 7/8 of the chip idle.  Reference lines: the Linear +
// dropout + residual + LayerNorm chains of Conformer.py:458-476,507,530-536 and summary_mixing.py:282-284.
//
// Lane / chunk layout and the reduction trees are those of layernorm_fwd_fast / layernorm_fwd_pair_fast (rowwise.hip): lane l of
// chunk i owns columns (l + 64 i) * 4 .. + 3; the statistics equal the standalone kernels' to an ulp.
#include "smx_common.h"

namespace smx {

struct SlabEpiParams {
  const float* slabs; long slab_stride; int nslab;
  int N, M;
  const float* bias;
  int act;
  void* Z; long ldz;                       // saved pre-activation (dtype T), or null
  uint32_t dthresh; float dscale; uint64_t dseed; const uint64_t* epoch; int drop_cols;
  float alpha; const uint8_t* row_mask;
  const void* res; long ldr; int res_f32;  // residual: float32 stream or dtype T
  void* C; long ldc; int c_f32;            // output: float32 stream or dtype T
  int ln;                                  // 1: LayerNorm appended
  const float* g1; const float* b1; float eps1; int lact; void* Y; long ldy; int y_f32; float* stats1;
  const float* g2; const float* b2; float eps2; void* Y2; long ldy2; float* stats2;   // second LayerNorm (of Y), Y2 dtype T; g2 null: none
};

template <int CH, int U>
__global__ __launch_bounds__(256) void slab_epilogue_kernel(const SlabEpiParams p) {
  typedef bf16_t T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = p.M;
  const uint64_t dseed = p.dthresh ? epoch_seed(p.dseed, p.epoch) : 0;
  const float invD = 1.f / (float)D;
  auto row_sum = [&](float (&v)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] += __shfl_xor(v[u], off, 64);
  };
  dispatch_act(p.act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    for (int row0 = (blockIdx.x * 4 + w) * U; row0 < p.N; row0 += gridDim.x * 4 * U) {
      float f[U][CH][4];
      // ---- the slabs of all U rows first (every load in flight before the first sum), then the residual ----
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + u, p.N - 1);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * 4;
          f[u][i][0] = f[u][i][1] = f[u][i][2] = f[u][i][3] = 0.f;
          if (c < D) {
            const float* sp = p.slabs + (long)row * D + c;
            for (int s0 = 0; s0 < p.nslab; s0 += 4) {        // four slabs in flight; summed in slab order
              float4 a[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const float4*>(sp + (long)min(s0 + k, p.nslab - 1) * p.slab_stride);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (s0 + k < p.nslab) { f[u][i][0] += a[k].x; f[u][i][1] += a[k].y; f[u][i][2] += a[k].z; f[u][i][3] += a[k].w; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = min(row0 + u, p.N - 1);
        const bool live = row0 + u < p.N;
        const float mk = (p.row_mask ? (p.row_mask[row] ? 1.f : 0.f) : 1.f) * p.alpha;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = (lane + 64 * i) * 4;
          if (c >= D) continue;
          float v[4] = {f[u][i][0], f[u][i][1], f[u][i][2], f[u][i][3]};
          if (p.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + c);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (p.Z && live) store4<T>(reinterpret_cast<T*>(p.Z) + (long)row * p.ldz + c, v);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = act_fwd_c<ACT>(v[j]);
          if (p.dthresh && c < p.drop_cols) dropout_apply_any<4>(v, dseed, (uint64_t)row * p.drop_cols + c, p.dthresh, p.dscale);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= mk;
          if (p.res) {
            float r[4];
            if (p.res_f32) load4<float>(reinterpret_cast<const float*>(p.res) + (long)row * p.ldr + c, r);
            else load4<T>(reinterpret_cast<const T*>(p.res) + (long)row * p.ldr + c, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j];
          }
          if (live) {
            if (p.c_f32) store4<float>(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + c, v);
            else store4<T>(reinterpret_cast<T*>(p.C) + (long)row * p.ldc + c, v);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) f[u][i][j] = v[j];
        }
      }
      if (!p.ln) continue;                                 // (uniform)
      // ---- LayerNorm(s) of the rows in registers: pass 0 = LN1 (+ lact), pass 1 = LN2 of LN1's output ----
      float s[U], q[U];
      const int npass = p.g2 ? 2 : 1;
      for (int pass = 0; pass < npass; ++pass) {
        const float* gam = pass == 0 ? p.g1 : p.g2;
        const float* bet = pass == 0 ? p.b1 : p.b2;
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
          const bool live = row < p.N;
          const float rstd = rsqrtf(q[u] * invD + (pass == 0 ? p.eps1 : p.eps2));
          float* st = pass == 0 ? p.stats1 : p.stats2;
          if (live && st && lane == 0) *reinterpret_cast<float2*>(st + 2 * (long)row) = make_float2(s[u], rstd);
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c >= D) continue;
            const float4 g4 = *reinterpret_cast<const float4*>(gam + c), b4 = *reinterpret_cast<const float4*>(bet + c);
            float o[4] = {(f[u][i][0] - s[u]) * rstd * g4.x + b4.x, (f[u][i][1] - s[u]) * rstd * g4.y + b4.y,
                          (f[u][i][2] - s[u]) * rstd * g4.z + b4.z, (f[u][i][3] - s[u]) * rstd * g4.w + b4.w};
            if (pass == 0) {
              if (p.lact != SMX_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = act_fwd(p.lact, o[j]);
              }
              if (live) {
                if (p.y_f32) store4<float>(reinterpret_cast<float*>(p.Y) + (long)row * p.ldy + c, o);
                else store4<T>(reinterpret_cast<T*>(p.Y) + (long)row * p.ldy + c, o);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) f[u][i][j] = o[j];
            } else if (live) {
              store4<T>(reinterpret_cast<T*>(p.Y2) + (long)row * p.ldy2 + c, o);
            }
          }
        }
      }
    }
  });
}

}  // namespace smx

using namespace smx;

extern "C" int smx_slab_epilogue_ok(int dtype, int N, int M, int nslab) {
  return dtype == SMX_BF16 && N >= 1 && M >= 4 && M % 4 == 0 && M <= 1024 && nslab >= 1 && nslab <= 16;
}

extern "C" int smx_slab_epilogue(int dtype, const float* slabs, int nslab, int64_t slab_stride, void* C, int64_t ldc, int N, int M,
                                 const smx_epilogue* epi, void* stream) {
  SMX_REQUIRE(slabs && C && epi, "smx_slab_epilogue: null pointer");
  if (N <= 0) return SMX_OK;
  if (!smx_slab_epilogue_ok(dtype, N, M, nslab)) return fail(SMX_EUNSUPPORTED, "smx_slab_epilogue: bf16 model, M %% 4 == 0, M <= 1024, 1..16 slabs (smx_slab_epilogue_ok)");
  const smx_epilogue& e = *epi;
  if (e.c0 || e.c0_mode != SMX_C0_NONE || e.colsum || (e.flags & ~SMX_EPI_LN_FWD) || e.out_mode == SMX_OUT_ATOMIC_F32 ||
      (e.io_flags & SMX_IO_LNX_F32) || e.bias_batch_stride)
    return fail(SMX_EUNSUPPORTED, "smx_slab_epilogue: bias / activation + z / dropout / alpha / row_mask / res / SMX_EPI_LN_FWD (+ lnf2_*) only; use smx_gemm");
  SMX_REQUIRE(e.drop_p >= 0.f && e.drop_p < 1.f && e.drop_cols >= 0 && e.drop_cols <= M && e.drop_cols % 4 == 0, "smx_slab_epilogue: bad dropout spec");
  const bool ln = (e.flags & SMX_EPI_LN_FWD) != 0;
  if (ln) SMX_REQUIRE(e.lnf_gamma && e.lnf_beta && e.lnf_y, "smx_slab_epilogue: SMX_EPI_LN_FWD needs lnf_gamma / lnf_beta / lnf_y");
  if (e.lnf2_y) SMX_REQUIRE(ln && e.lnf2_gamma && e.lnf2_beta, "smx_slab_epilogue: lnf2_* goes with SMX_EPI_LN_FWD");
  const bool c_f32 = e.out_mode == SMX_OUT_F32, r_f32 = (e.io_flags & SMX_IO_RES_F32) != 0, y_f32 = (e.io_flags & SMX_IO_LNFY_F32) != 0;
  auto rows_ok = [&](const void* ptr, int64_t ld, bool f32) {   // 4-element vector accesses: 16-byte (float32) / 8-byte (bf16) aligned rows
    const int64_t a = f32 ? 16 : 8;
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) % a) == 0 && ld % 4 == 0 && ld >= M);
  };
  SMX_REQUIRE(aligned16(slabs) && slab_stride % 4 == 0 && rows_ok(C, ldc, c_f32) && rows_ok(e.z, e.ldz, false) && rows_ok(e.res, e.ldr, r_f32) &&
                  rows_ok(e.lnf_y, e.lnf_ldy, y_f32) && rows_ok(e.lnf2_y, e.lnf2_ldy, false) && (!e.bias || aligned16(e.bias)) &&
                  (!ln || (aligned16(e.lnf_gamma) && aligned16(e.lnf_beta))) && (!e.lnf2_y || (aligned16(e.lnf2_gamma) && aligned16(e.lnf2_beta))),
              "smx_slab_epilogue: rows must be vector aligned (ld %% 4 == 0, 16-byte float32 / 8-byte bf16 bases)");
  SlabEpiParams p;
  memset(&p, 0, sizeof(p));
  p.slabs = slabs; p.slab_stride = slab_stride; p.nslab = nslab; p.N = N; p.M = M;
  p.bias = e.bias; p.act = e.act; p.Z = e.z; p.ldz = e.ldz;
  p.dthresh = (uint32_t)((double)e.drop_p * 4294967296.0); p.dscale = 1.f / (1.f - e.drop_p); p.dseed = e.drop_seed; p.epoch = e.epoch;
  p.drop_cols = e.drop_cols > 0 ? e.drop_cols : M;
  p.alpha = e.alpha; p.row_mask = e.row_mask;
  p.res = e.res; p.ldr = e.ldr; p.res_f32 = r_f32;
  p.C = C; p.ldc = ldc; p.c_f32 = c_f32;
  p.ln = ln;
  p.g1 = e.lnf_gamma; p.b1 = e.lnf_beta; p.eps1 = e.lnf_eps; p.lact = e.lnf_act; p.Y = e.lnf_y; p.ldy = e.lnf_ldy; p.y_f32 = y_f32; p.stats1 = e.lnf_stats;
  p.g2 = e.lnf2_y ? e.lnf2_gamma : nullptr; p.b2 = e.lnf2_beta; p.eps2 = e.lnf2_eps; p.Y2 = e.lnf2_y; p.ldy2 = e.lnf2_ldy; p.stats2 = e.lnf2_stats;
  const int ch = (M + 255) / 256;
  const int U = ch <= 1 ? 2 : 1;                           // rows in flight per wave (the slabs multiply the loads per row)
  long blocks = ((long)N + 4 * U - 1) / (4 * U);
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SMX_SE(CH_, U_) hipLaunchKernelGGL((slab_epilogue_kernel<CH_, U_>), dim3((unsigned)blocks), dim3(256), 0, s, p)
  if (ch <= 1) SMX_SE(1, 2);
  else if (ch <= 2) SMX_SE(2, 1);
  else if (ch <= 3) SMX_SE(3, 1);
  else SMX_SE(4, 1);
#undef SMX_SE
  return check_launch("smx_slab_epilogue");
}
