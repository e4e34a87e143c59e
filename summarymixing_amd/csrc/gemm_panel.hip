// gemm_panel.hip — C-ABI of the panel-resident GEMM (gemm_panel.h): smx_weight_pack, smx_gemm_panel, the forward instantiations.
// (The act-grad instantiations live in gemm_panel_bwd.hip: a translation unit of its own, compiled next to this one.)
#include "gemm_panel.h"

namespace smx {

// One wave per 1 KB fragment: 32 output columns x 16 reduce elements in MFMA operand order - lane (c = lane & 31, hi = lane >> 5)
// holds B[m = cb * 32 + c][k = kk * 16 + hi * 8 .. + 8] - stored lane-major at ((cb * (KS + 1) + kk) * 64 + lane) * 16 bytes.
// transposed = 0: W is (M, K), reduce-contiguous (a Linear's weight in its forward);
// transposed = 1: W is (K, M), the same Linear's weight seen from its dgrad (B[m][k] = W[k][m]).
// Fragment kk = KS of a column block is the bias: reduce slot 0 = bf16(b), slot 1 = bf16(b - slot 0), the rest 0 (zeros without a bias).
__global__ __launch_bounds__(256) void weight_pack_kernel(const uint16_t* __restrict__ W, long ldw, int transposed, const float* __restrict__ bias,
                                                          int M, int K, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hi = lane >> 5;
  const int KS = K >> 4;
  const long frag = (long)blockIdx.x * 4 + wave;          // = cb * (KS + 1) + kk  (4 consecutive kk per block: 128-byte row segments)
  if (frag >= (long)(M >> 5) * (KS + 1)) return;
  const int cb = (int)(frag / (KS + 1)), kk = (int)(frag % (KS + 1));
  const int m = cb * 32 + c, k0 = kk * 16 + hi * 8;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (kk == KS) {
    if (bias && hi == 0) {
      const float b = bias[m];
      const uint32_t bh = f32_to_bf16_bits(b);
      v.x = bh | (f32_to_bf16_bits(b - bf16_bits_to_f32(bh)) << 16);
    }
  } else if (!transposed) {
    v = *reinterpret_cast<const uint4*>(W + (long)m * ldw + k0);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      w[q] = (uint32_t)W[(long)(k0 + 2 * q) * ldw + m] | ((uint32_t)W[(long)(k0 + 2 * q + 1) * ldw + m] << 16);
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  out[frag * 64 + lane] = v;
}

// the same for a table of weights in ONE launch (every packed image of a model after an optimizer step): block b belongs to the
// last job whose block_start <= b
__global__ __launch_bounds__(256) void weight_pack_jobs_kernel(const smx_pack_job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const smx_pack_job j = jobs[lo];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, hi_ = lane >> 5;
  const int KS = j.K >> 4;
  const long frag = (long)((int)blockIdx.x - j.block_start) * 4 + wave;
  if (frag >= (long)(j.M >> 5) * (KS + 1)) return;
  const int cb = (int)(frag / (KS + 1)), kk = (int)(frag % (KS + 1));
  const int m = cb * 32 + c, k0 = kk * 16 + hi_ * 8;
  const uint16_t* W = reinterpret_cast<const uint16_t*>(j.W);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (kk == KS) {
    if (j.bias && hi_ == 0) {
      const float b = j.bias[m];
      const uint32_t bh = f32_to_bf16_bits(b);
      v.x = bh | (f32_to_bf16_bits(b - bf16_bits_to_f32(bh)) << 16);
    }
  } else if (!j.transposed) {
    v = *reinterpret_cast<const uint4*>(W + (long)m * j.ldw + k0);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      w[q] = (uint32_t)W[(long)(k0 + 2 * q) * j.ldw + m] | ((uint32_t)W[(long)(k0 + 2 * q + 1) * j.ldw + m] << 16);
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  reinterpret_cast<uint4*>(j.packed)[frag * 64 + lane] = v;
}

int launch_panel_fwd(const PanelParams& p, int K, int act, hipStream_t s) { return launch_panel_mode<0>(p, K, act, s); }

}  // namespace smx

using namespace smx;
#ifdef SMX_DIAG
extern long long* g_dbg_stamps;   // gemm.hip (smx_debug_set_timing_buffer)
#endif

// one workgroup per CU: with fewer panels than ~3/4 of the CUs a panel's chunk rounds (M / 512 of them) are dealt to 2 or 4 workgroups,
// and when that still leaves most of the chip idle (a few thousand frames) the panels shrink to 64 or 32 rows (SMX_PANEL_ROWS=<n>: that size)
static void panel_geometry(int N, int M, int* rows_out, int* csplit_out) {
  const int rounds = (M / 64 + 7) / 8;
  auto split_for = [&](int rows) {
    const int panels = (N + rows - 1) / rows;
    int cs = 1;
    while (cs < 4 && panels * cs < 192 && rounds % (cs * 2) == 0) cs *= 2;
    return cs;
  };
  int rows = 128;
  const int forced = cfg().panel_rows;
  if (forced == 128 || forced == 64 || forced == 32) rows = forced;
  else {
    // the LARGEST panel height whose launch still has ~3/4 of a workgroup per CU (tools/experiments/r06_smalln: one round of
    // 188-256 workgroups is the best point of every (frames, M) cell between 2 000 and 12 000 frames)
    while (rows > 32 && ((N + rows - 1) / rows) * split_for(rows) < 180) rows >>= 1;
  }
  int cs = split_for(rows);
  // One panel per CU at a time, so the launch runs in rounds of 256 workgroups and a mostly empty last round costs a whole one: C2b
  // B = 72 x 500 is 282 panels of 128 rows - the up-projection took 54 us against 33 us at B = 64 (250 panels).  Among 128- / 64-row panels
  // with the chunk rounds dealt to 1, 2 or 4 workgroups, take the geometry that fills its rounds best (a 64-row panel streams the
  // weights twice per 128 rows: -5 %; every doubling of the split stages the panel once more: -2 %).
  // Up to four rounds only: beyond, the last round is a small part of the launch and the panel staged twice comes from HBM, not L2
  // (config 5, 240 000 rows = 1875 panels: 484 -> 534 us and 308 -> 363 us with the chunk rounds dealt to two workgroups).
  if (forced == 0 && rows == 128 && (long)((N + 127) / 128) * cs > 256 && (long)((N + 127) / 128) * cs <= 1024) {
    double best = -1.0;
    int brows = 128, bcs = cs;
    for (int r = 128; r >= 64; r >>= 1) {
      for (int c = cs; c <= 4; c *= 2) {
        if (rounds % c != 0) break;
        const long wgs = (long)((N + r - 1) / r) * c;
        double score = (double)wgs / (256.0 * (double)((wgs + 255) / 256));
        if (r == 64) score *= 0.95;
        for (int q = cs; q < c; q *= 2) score *= 0.98;
        if (score > best + 1e-9) { best = score; brows = r; bcs = c; }
      }
    }
    rows = brows; cs = bcs;
  }
  *rows_out = rows;
  *csplit_out = cs;
}

extern "C" int smx_gemm_panel_rows(int N, int M) {
  int rows = 128, cs = 1;
  if (N > 0 && M > 0) panel_geometry(N, M, &rows, &cs);
  return rows;
}

extern "C" int smx_gemm_panel_ok(int dtype, int N, int M, int K) {
  return dtype == SMX_BF16 && (K == 256 || K == 512) && N >= 1 && M >= 64 && M % 64 == 0 && (long)N * M * 2 < (1L << 31);
}

extern "C" size_t smx_weight_pack_bytes(int M, int K) { return (M > 0 && K > 0) ? (size_t)M * (size_t)(K + 16) * 2 : 0; }

extern "C" int smx_weight_pack(int dtype, const void* W, int64_t ldw, int transposed, const float* bias, int M, int K, void* packed, void* stream) {
  SMX_REQUIRE(W && packed, "smx_weight_pack: null pointer");
  SMX_REQUIRE(dtype == SMX_BF16, "smx_weight_pack: bf16 only");
  SMX_REQUIRE(M > 0 && K > 0 && M % 32 == 0 && K % 16 == 0, "smx_weight_pack: M %% 32 == 0 and K %% 16 == 0 (got M=%d K=%d)", M, K);
  SMX_REQUIRE(aligned16(packed), "smx_weight_pack: the packed image must be 16-byte aligned");
  if (!transposed) SMX_REQUIRE(aligned16(W) && ldw % 8 == 0 && ldw >= K, "smx_weight_pack: (M, K) weight rows must be 16-byte aligned");
  else SMX_REQUIRE(ldw >= M, "smx_weight_pack: bad leading dimension");
  const long nfrag = (long)(M / 32) * (K / 16 + 1);
  hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(W), (long)ldw, transposed, bias, M, K, reinterpret_cast<uint4*>(packed));
  return check_launch("smx_weight_pack");
}

extern "C" int smx_weight_pack_job_blocks(int M, int K) {
  return (M > 0 && K > 0) ? (int)(((long)(M / 32) * (K / 16 + 1) + 3) / 4) : 0;
}

extern "C" int smx_weight_pack_jobs(int dtype, const smx_pack_job* jobs_dev, int njobs, int total_blocks, void* stream) {
  SMX_REQUIRE(dtype == SMX_BF16, "smx_weight_pack_jobs: bf16 only");
  SMX_REQUIRE(njobs >= 0 && total_blocks >= 0, "smx_weight_pack_jobs: bad sizes");
  if (njobs == 0 || total_blocks == 0) return SMX_OK;
  SMX_REQUIRE(jobs_dev, "smx_weight_pack_jobs: null pointer");
  hipLaunchKernelGGL(weight_pack_jobs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev, njobs);
  return check_launch("smx_weight_pack_jobs");
}

extern "C" int smx_gemm_panel(int dtype, const void* A, int64_t lda, const void* Wpacked, void* C, int64_t ldc, int N, int M, int K,
                              const smx_epilogue* epi, void* stream) {
  SMX_REQUIRE(A && Wpacked && C, "smx_gemm_panel: null operand");
  SMX_REQUIRE(N >= 0 && M >= 0 && K >= 0, "smx_gemm_panel: bad sizes N=%d M=%d K=%d", N, M, K);
  if (N == 0 || M == 0) return SMX_OK;
  if (!smx_gemm_panel_ok(dtype, N, M, K)) return fail(SMX_EUNSUPPORTED, "smx_gemm_panel: bf16, K = 256 / 512, M %% 64 == 0, output < 2 GB (smx_gemm_panel_ok)");
  smx_epilogue e;
  if (epi) e = *epi;
  else { memset(&e, 0, sizeof(e)); e.alpha = 1.f; }
  const bool ag = (e.flags & SMX_EPI_ACT_GRAD) != 0;
  if (e.c0 || e.c0_mode != SMX_C0_NONE || e.res || e.colsum || e.out_mode != SMX_OUT_T ||
      (e.flags & ~SMX_EPI_ACT_GRAD) || e.io_flags || e.lnf2_y || e.bias ||
      !(e.act == SMX_ACT_NONE || e.act == SMX_ACT_SWISH || e.act == SMX_ACT_GELU || e.act == SMX_ACT_RELU) ||
      e.drop_cols < 0 || e.drop_cols > M || e.drop_cols % 64 != 0)
    return fail(SMX_EUNSUPPORTED, "smx_gemm_panel: epilogue = activation (none / Swish / GELU / ReLU), saved Z, dropout, row mask, alpha, or SMX_EPI_ACT_GRAD (the bias belongs to smx_weight_pack); use smx_gemm");
  SMX_REQUIRE(!ag || e.z, "smx_gemm_panel: SMX_EPI_ACT_GRAD needs z (input)");
  SMX_REQUIRE(e.drop_p >= 0.f && e.drop_p < 1.f, "smx_gemm_panel: 0 <= drop_p < 1");
  SMX_REQUIRE(aligned16(A) && lda % 8 == 0 && lda >= K && aligned16(Wpacked) && aligned16(C) && ldc % 8 == 0 && ldc >= M &&
                  (!e.z || (aligned16(e.z) && e.ldz % 8 == 0 && e.ldz >= M)),
              "smx_gemm_panel: operands must be 16-byte aligned with leading dimensions %% 8 == 0");
  // (32-bit buffer offsets: every operand - with ITS leading dimension, a column slice of a wider buffer counts in full - below 2 GB)
  if (((long)N - 1) * lda * 2 + (long)K * 2 >= (1L << 31) || ((long)N - 1) * ldc * 2 + (long)M * 2 >= (1L << 31) ||
      (e.z && ((long)N - 1) * e.ldz * 2 + (long)M * 2 >= (1L << 31)))
    return fail(SMX_EUNSUPPORTED, "smx_gemm_panel: operand spans (rows x leading dimension) must stay below 2 GB; use smx_gemm");
  PanelParams p;
  memset(&p, 0, sizeof(p));
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda;
  p.Bp = Wpacked;
  p.C = reinterpret_cast<bf16_t*>(C); p.ldc = ldc;
  p.Z = reinterpret_cast<bf16_t*>(e.z); p.ldz = e.ldz;
  p.N = N; p.M = M;
  p.dthresh = (unsigned)((double)e.drop_p * 4294967296.0);
  p.dscale = 1.f / (1.f - e.drop_p);
  p.seed = e.drop_seed; p.epoch = e.epoch;
  p.row_mask = e.row_mask; p.alpha = e.alpha;
  p.drop_cols = e.drop_cols > 0 ? e.drop_cols : M;
  // store policy of smx_gemm: the output is streamed once it cannot survive in the Infinity Cache anyway
  p.nt = ((long)N * M * 2 >= (96L << 20)) ? 2 : 0;
  panel_geometry(N, M, &p.rows, &p.csplit);
#ifdef SMX_DIAG
  p.dbg = g_dbg_stamps;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return ag ? launch_panel_actgrad(p, K, e.act, s) : launch_panel_fwd(p, K, e.act, s);
}

// ---- split-K over workgroups for the LONG reductions of a small batch (round 6) -----------------------------------------------------
// C (N x M) = A (N x nslice K) W^T with the reduction cut into nslice panel-sized slices: workgroup (panel, slice) multiplies its
// K-slice on the panel-resident kernel (MODE 2) and stores float32 partial products to slab[slice]; smx_slab_epilogue (rowwise.hip)
// adds the slabs in a fixed order and applies the Linear's epilogue - and the LayerNorm that follows it.  For the recipe's 3750
// frames the 64 x 64 tiles of smx_gemm walk K = 2048 on 472 LDS-bound workgroups (21-25 us); 59 panels x 4 slices = 236 workgroups of
// 8 waves do the same work with the weights never in LDS.
extern "C" int smx_gemm_panel_slabs_ok(int dtype, int N, int M, int K, int nslice) {
  return dtype == SMX_BF16 && (K == 256 || K == 512) && N >= 1 && M >= 64 && M % 64 == 0 && M <= 512 && nslice >= 1 && nslice <= 16 &&
         (long)N * M * 4 < (1L << 31) && (long)N * nslice * K * 2 < (1L << 31);
}

extern "C" int smx_gemm_panel_slabs(int dtype, const void* A, int64_t lda, const void* Wpacked, float* slabs, int N, int M, int K, int nslice,
                                    void* stream) {
  SMX_REQUIRE(A && Wpacked && slabs, "smx_gemm_panel_slabs: null operand");
  if (N <= 0) return SMX_OK;
  if (!smx_gemm_panel_slabs_ok(dtype, N, M, K, nslice)) return fail(SMX_EUNSUPPORTED, "smx_gemm_panel_slabs: bf16, K = 256 / 512 per slice, M %% 64 == 0, M <= 512 (smx_gemm_panel_slabs_ok)");
  SMX_REQUIRE(aligned16(A) && lda % 8 == 0 && lda >= (int64_t)nslice * K && aligned16(Wpacked) && aligned16(slabs),
              "smx_gemm_panel_slabs: operands must be 16-byte aligned with lda %% 8 == 0 and lda >= nslice * K");
  if (((long)N - 1) * lda * 2 + (long)nslice * K * 2 >= (1L << 31)) return fail(SMX_EUNSUPPORTED, "smx_gemm_panel_slabs: operand span must stay below 2 GB");
  PanelParams p;
  memset(&p, 0, sizeof(p));
  p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda;
  p.Bp = Wpacked;
  p.N = N; p.M = M;
  p.alpha = 1.f; p.dscale = 1.f; p.drop_cols = M;
  p.slab = slabs; p.slab_stride = (long)N * M; p.nslice = nslice;
  p.b_slice_bytes = (long)smx_weight_pack_bytes(M, K);
  {
    // panel height: the largest with ~3/4 of a workgroup per CU (cf. smx_gemm_panel); M <= 512 is at most one chunk round: csplit = 1
    int rows = K == 512 ? 64 : 128;                        // (K = 512: the float32 scratch leaves room for 64 rows)
    const int forced = cfg().panel_rows;
    if (forced == 64 || forced == 32 || (forced == 128 && K == 256)) rows = forced;
    else while (rows > 32 && (long)((N + rows - 1) / rows) * nslice < 180) rows >>= 1;
    p.rows = rows;
    p.csplit = 1;
  }
#ifdef SMX_DIAG
  p.dbg = g_dbg_stamps;
#endif
  return launch_panel_slabs(p, K, reinterpret_cast<hipStream_t>(stream));
}
