/* smx.h — C-ABI of libsmx.so: the MI355X (gfx950) SummaryMixing encoder hot path.
 *
 * The reference (SamsungLabs/SummaryMixing @ 2024_10_08) is pure Python on torch and has NO FFI and no
 * native kernel: every entry point below is NEW.  Each one names the reference lines whose arithmetic it
 * replaces (paths relative to the reference root) so that a maintainer can bind it from the reference's
 * nn.Modules (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *  - every function returns SMX_OK (0) or a negative SMX_E* code; never throws, never aborts;
 *    smx_last_error() returns a thread-local message for the last failing call on this thread.
 *  - all data pointers are DEVICE pointers owned by the caller; the library allocates nothing.
 *  - matrices are row-major with an explicit leading dimension (in ELEMENTS) so column slices of a
 *    wider buffer are addressed without copies.
 *  - dtype is SMX_F32 or SMX_BF16 for activations/weights ("T"); biases, LayerNorm affine, conv taps,
 *    per-utterance side inputs, statistics and all gradients of parameters are always float32.
 *  - last argument is the hipStream_t to launch on (passed as void*); all work is stream ordered and
 *    asynchronous; no global mutable state; safe from several host threads on different streams.
 */
#ifndef SMX_H_
#define SMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_VERSION 100

enum { SMX_OK = 0, SMX_EINVAL = -1, SMX_EUNSUPPORTED = -2, SMX_ELAUNCH = -3 };
enum { SMX_F32 = 0, SMX_BF16 = 1 };
enum { SMX_ACT_NONE = 0, SMX_ACT_GELU = 1, SMX_ACT_SWISH = 2, SMX_ACT_LEAKY_RELU = 3, SMX_ACT_RELU = 4 };
/* how the fp32 side input C0 is indexed by output row n: none | row n | row n / div | row n % div */
enum { SMX_C0_NONE = 0, SMX_C0_ROW = 1, SMX_C0_GROUP = 2, SMX_C0_MOD = 3 };
/* operand storage for smx_gemm: KC = reduce dimension contiguous, KS = reduce dimension strided */
enum { SMX_GEMM_NT = 0, /* A (N,K) KC, B (M,K) KC : Y = X W^T        (torch Linear forward)            */
       SMX_GEMM_NN = 1, /* A (N,K) KC, B (K,M) KS : dX = dZ W        (dgrad; ParallelLinear einsum)     */
       SMX_GEMM_TN = 2  /* A (K,N) KS, B (K,M) KS : dW = dZ^T X      (wgrad)                            */ };
enum { SMX_OUT_T = 0, SMX_OUT_F32 = 1, SMX_OUT_ATOMIC_F32 = 2 };
enum { SMX_PAD_ZERO = 0, SMX_PAD_REFLECT = 1 };

int smx_version(void);
const char* smx_last_error(void);

/* The library's tuning / diagnostic knobs.  They are read from the environment ONCE, when the library is first used, into
 * this struct (never again: no getenv on any call path); smx_get_config returns the values in force.  Every field defaults
 * to the measured-best setting; the SMX_* variable that overrides it is named in the comment.  Ablation switches
 * (SMX_GEMM_ABLATE, SMX_WGROUP_ABLATE, SMX_DWROLL_ABLATE) only exist in builds with -DSMX_DIAG and read 0 otherwise. */
typedef struct smx_config {
  int32_t ln_tile_rows;     /* rows of the LayerNorm-fused GEMM tile (128; see smx_gemm_ln_tile_rows)        */
  int32_t gemm_ablate, wgroup_ablate, dwroll_ablate;   /* SMX_DIAG builds only                                  */
  int32_t diag_build;       /* 1 when the library was compiled with -DSMX_DIAG                               */
  int32_t t256;             /* SMX_T256: 256 x 256 GEMM tile (one workgroup per CU, software-pipelined K loop)
                               0 off, 1 for K >= 2048 (1), 2 for every eligible shape (tests)                  */
  int32_t panel_rows;       /* SMX_PANEL_ROWS: rows per panel of smx_gemm_panel (128 / 64 / 32); 0 = by frame count  */
  int32_t pool_fuse_max_rows;  /* SMX_POOL_FUSE_MAX_ROWS: smx_pool_bcast_ok up to this many frames (B * T)            */
  int32_t ln_tile64;        /* SMX_LN_TILE64: the 64 x 256 LayerNorm-fused tile: 0 never, 1 where it fills its rounds better (default), 2 always */
  int32_t pad_;
} smx_config;
int smx_get_config(smx_config* out);
/* rows per tile of the LayerNorm-fused GEMMs (SMX_EPI_LN_BWD writes ceil(N / rows) partial row pairs into ln_partial) */
int smx_gemm_ln_tile_rows(void);
/* ... for a given launch: 64 (d_model = 256 at mid-size batches, round 6) or 128 */
int smx_gemm_ln_tile_rows_for(int N, int M);

/* Epilogue of the fused projection GEMM:
 *   v      = acc + bias[m] + C0[map(n), m]
 *   Z[n,m] = v                                   (optional pre-activation store, dtype T)
 *   a      = dropout(act(v)) * row_mask[n]        (dropout: keep(seed, n*M+m) ? x/(1-p) : 0; off when drop_p == 0)
 *   C[n,m] = R[n,m] + alpha * a                  (R optional residual, dtype T)
 * With out_mode SMX_OUT_ATOMIC_F32: C (float32) += alpha * acc and every other epilogue field must be 0. */
typedef struct smx_epilogue {
  const float* bias;   int64_t bias_batch_stride;       /* [M] fp32 or NULL                           */
  const float* c0;     int64_t ldc0; int32_t c0_mode; int32_t c0_div;
  int32_t act;         int32_t out_mode;
  void* z;             int64_t ldz;
  const uint8_t* row_mask;                               /* [N] 1 = valid frame, or NULL               */
  const void* res;     int64_t ldr;
  float alpha;         int32_t flags;                     /* SMX_EPI_C0_POST: add C0 after dropout/mask/alpha      */
  float drop_p;        int32_t drop_cols;                 /* fused inverted dropout (0 = off), applied after act();
                                                           * drop_cols > 0: only the first drop_cols output columns,
                                                           * masks indexed n*drop_cols + m (0: all M, n*M + m)        */
  uint64_t drop_seed;                                     /* counter-based mask, same indexing as smx_dropout       */
  float* colsum;                                          /* [M] fp32 or NULL: colsum[m] += sum_n C[n,m] (fixed order) */
  void* workspace;                                        /* smx_gemm_colsum_workspace(N, M) bytes when colsum is set  */
  /* SMX_EPI_LN_BWD: the GEMM output is the gradient of a LayerNorm's OUTPUT (M = the LayerNorm width) */
  const void* ln_x;    int64_t ln_ldx;                    /* the LayerNorm input (N, M), dtype T                       */
  const float* ln_stats; const float* ln_gamma;           /* (N, 2) mean | rstd of the forward; [M]                    */
  float* ln_partial;                                      /* [ceil(N/smx_gemm_ln_tile_rows())][2][M] per-tile dgamma | dbeta partial rows */
  void* ln_dx2;        int64_t ln_lddx2;                  /* optional second output (see smx_layernorm_bwd2) or NULL   */
  const uint8_t* ln_mask2; float ln_alpha2; float ln_drop_p2; uint64_t ln_drop_seed2;
  /* SMX_EPI_LN_FWD: a LayerNorm of the (row-complete) output: lnf_y = act(LN(C)), lnf_stats = (mean, rstd) per row */
  const float* lnf_gamma; const float* lnf_beta; void* lnf_y; int64_t lnf_ldy; float* lnf_stats; float lnf_eps; int32_t lnf_act;
  /* fp32 residual stream (torch autocast semantics: Linear I/O in bf16, the residual adds and LayerNorm inputs in fp32 -
   * Conformer.py:507,530,532-536): SMX_IO_RES_F32 = `res` is float32 (needs out_mode SMX_OUT_F32: C is the new stream
   * tensor); SMX_IO_LNX_F32 = `ln_x` (SMX_EPI_LN_BWD) is float32.  lnf_y is dtype T (the next GEMM's input) unless
   * SMX_IO_LNFY_F32 (the LayerNorm output is itself the stream: the layer-final norm2, Conformer.py:536). */
  int32_t io_flags;    int32_t pad_;
  /* device step counter mixed into this call's fused dropout seed (see smx_step_counter_add), or NULL */
  const uint64_t* epoch;
  /* SMX_EPI_LN_FWD, optional SECOND LayerNorm of the first one's output (a Conformer layer's norm2 followed by the next layer's
   * first LayerNorm, Conformer.py:536 / :507): lnf2_y (dtype T) = LN2(lnf_y values before rounding), lnf2_stats = (mean, rstd).
   * Only where smx_gemm_ln_pair_ok says so (the 128 x 512 tile on the float32 stream); NULL lnf2_y = off. */
  const float* lnf2_gamma; const float* lnf2_beta; void* lnf2_y; int64_t lnf2_ldy; float* lnf2_stats; float lnf2_eps; int32_t pad2_;
} smx_epilogue;
enum { SMX_IO_RES_F32 = 1, SMX_IO_LNX_F32 = 2, SMX_IO_LNFY_F32 = 4 };
/* flags.  SMX_EPI_ACT_GRAD turns the epilogue into the BACKWARD of an upstream activation layer: z is then a
 * read-only INPUT (the pre-activation the forward saved) and
 *   C[n,m] = alpha * dropout(v * act'(z[n,m])) * row_mask[n]
 * so a dgrad GEMM dX = dZ W emits the upstream layer's dZ directly (act/dropout/mask backward fused; `res` must be
 * NULL); with `colsum` the upstream bias gradient comes out of the same launch.  batch == 1, splits == 1. */
/* SMX_EPI_LN_BWD fuses the LayerNorm BACKWARD into the dgrad GEMM that produces the gradient g of the LayerNorm's output
 * (torch.nn.LayerNorm of the Conformer modules, Conformer.py:146,152,458,475): with xhat = (ln_x - mean) * rstd,
 *   C[n,:] = rstd * (g*gamma - mean_m(g*gamma) - xhat * mean_m(g*gamma*xhat)) + res[n,:]     (res = residual gradient)
 * and per-tile partial rows dgamma = sum_n g * xhat, dbeta = sum_n g in ln_partial (fold with smx_reduce_jobs: two jobs,
 * src = ln_partial (+ M), src_stride 2*M, nsrc = ceil(N/128), rows 1, cols M).  Needs the whole LayerNorm row in one
 * tile: dtype bf16, M == 256, aligned operands, N >= 128 (smx_gemm_ln_fused_ok); no bias / dropout / C0.  With
 * lnf_act != 0 the LayerNorm had a fused activation (Y = act(LN(x)), lnf_beta required): g is first multiplied by
 * act'(LN(x)).  With z / act set (and ln_dx2) the second output is alpha2 * D(dX * act'(z)) * mask2 (the consumer's own
 * activation backward).
 * SMX_EPI_LN_FWD (same shape conditions) appends a LayerNorm FORWARD of the finished output rows: the ordinary epilogue
 * writes C (bias, residual, dropout, mask ... as usual), then lnf_y = act(LN(C) * gamma + beta) and lnf_stats. */
enum { SMX_EPI_C0_POST = 1, SMX_EPI_ACT_GRAD = 2, SMX_EPI_LN_BWD = 4, SMX_EPI_LN_FWD = 8 };
int smx_gemm_ln_fused_ok(int dtype, int N, int M, int K);
/* ... and can that epilogue run a SECOND LayerNorm on the first one's output (smx_epilogue.lnf2_*)?  Needs, besides the shape, the
 * float32 residual stream form of the call: SMX_OUT_F32, SMX_IO_RES_F32 residual, no C0 / column sums (else SMX_EUNSUPPORTED). */
int smx_gemm_ln_pair_ok(int dtype, int N, int M, int K);
size_t smx_gemm_colsum_workspace(int N, int M);

/* Batched strided MFMA GEMM  C[b] (N x M) = epilogue( op(A[b]) . op(B[b]) ), reduce length K.
 * Replaces: nn.Linear inside VanillaNN (VanillaNN.py:189-196, summary_mixing.py:207,210,237,257,282),
 * the per-head einsum of ParallelLinear (VanillaNN.py:108-112; batch = n_split), the FFN / pointwise-conv /
 * post-conv Linears of the Conformer layer (Conformer.py:128-157,458-472) and their autograd backward.
 * splits > 1 partitions K over blockIdx (requires out_mode SMX_OUT_ATOMIC_F32). */
int smx_gemm(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb,
             int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K, int batch, int splits,
             const smx_epilogue* epi, void* stream);
/* Which kernel would smx_gemm launch for these arguments?  Same checks, same dispatch, no launch: fills `plan` with the template
 * arguments of the instantiation as a profiler prints them - gemm_kernel<T, a_kc, b_kc, tile_n, tile_m, vec, lnf, gather>
 * (kernel 0) or gemm_tn_dma_kernel (kernel 1).  bench.py groups its in-step records by this symbol. */
typedef struct smx_gemm_plan { int32_t kernel, a_kc, b_kc, tile_n, tile_m, vec, lnf, gather; } smx_gemm_plan;
int smx_gemm_plan_query(int layout, int dtype, const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb,
                        int64_t strideB, void* C, int64_t ldc, int64_t strideC, int N, int M, int K, int batch, int splits,
                        const smx_epilogue* epi, smx_gemm_plan* plan);

/* Panel-resident GEMM for the SHORT reductions with WIDE outputs of an encoder layer (bf16, K = 256 or 512, M % 64 == 0):
 *   forward      C = alpha * D(act(A Wp + bias)) * row_mask, optionally saving the pre-activation Z   (epi: act, z, drop_* [drop_cols % 64 == 0], row_mask, alpha; bias: packed, below)
 *   act-grad     C = alpha * D((A Wp) * act'(Z)) * row_mask                                       (epi: SMX_EPI_ACT_GRAD, z = input, act, drop_*, row_mask, alpha)
 * i.e. the FFN up-projection  nn.Linear(d_model, d_ffn) + activation + dropout  (Conformer.py:458-472, Branchformer.py:142-157)
 * and the first half of the autograd backward of the down-projection that follows it (dH = dY W2, then the activation / dropout
 * backward), the two output-bound GEMMs of every encoder layer.  A 128-row panel of A stays in LDS for all M columns and
 * the weight is read in MFMA fragment order straight into registers, so it must be PRE-PACKED:
 *   smx_weight_pack(W, transposed = 0, bias): W is the (M, K) weight of the forward, bias its fp32 [M] bias or NULL;
 *   smx_weight_pack(W, transposed = 1, NULL): W is the (K, M) weight of the Linear whose dgrad this is (dH = dY W: reduce-strided).
 * The bias travels INSIDE the packed image (one more fragment per 32 columns, split into a bf16 high and low part: exact to
 * 2^-17 relative), so smx_gemm_panel takes no epi->bias.  The image has smx_weight_pack_bytes(M, K) bytes (16-byte aligned) and
 * must be re-packed whenever W or the bias change.
 * Same arithmetic as smx_gemm with the same epilogue fields (fp32 accumulation, the same dropout mask for the same seed), except
 * that the activation / its gradient is evaluated on the bf16-ROUNDED pre-activation / product (torch autocast semantics: the
 * Linear's output is a bf16 tensor).  Any other epilogue field (bias, residual, C0, LayerNorm, fp32 output,
 * column sums) returns SMX_EUNSUPPORTED: use smx_gemm.  smx_gemm_panel_ok: can these sizes take the panel path at all? */
int smx_gemm_panel_ok(int dtype, int N, int M, int K);
/* rows per panel the launch for (N, M) would use: 128, or 64 / 32 when a small batch has too few 128-row panels for the chip (round 6) */
int smx_gemm_panel_rows(int N, int M);
size_t smx_weight_pack_bytes(int M, int K);
int smx_weight_pack(int dtype, const void* W, int64_t ldw, int transposed, const float* bias, int M, int K, void* packed, void* stream);
int smx_gemm_panel(int dtype, const void* A, int64_t lda, const void* Wpacked, void* C, int64_t ldc, int N, int M, int K,
                   const smx_epilogue* epi, void* stream);
/* Many smx_weight_pack calls in ONE launch (every packed image of a model, right after the optimizer rewrote the weights): a
 * table of jobs in DEVICE memory with the same per-job requirements as smx_weight_pack; block_start = the prefix sum of
 * smx_weight_pack_job_blocks(M, K) over the jobs before this one, total_blocks = the sum over all of them. */
typedef struct smx_pack_job {
  const void* W; int64_t ldw; const float* bias; void* packed; int32_t M, K, transposed, block_start;
} smx_pack_job;
int smx_weight_pack_job_blocks(int M, int K);
int smx_weight_pack_jobs(int dtype, const smx_pack_job* jobs_dev, int njobs, int total_blocks, void* stream);

/* Weight gradient of a (batched) Linear: dW[b] (M x K) += alpha * dZ[b]^T X[b], reducing over `rows` frames
 * (dZ (rows, M), X (rows, K), both row-major).  Split-K over the frame dimension into fp32 slabs in `workspace`
 * (smx_linear_wgrad_workspace bytes) followed by one fixed-order reduction: bit-reproducible, no atomics.
 * dbias (optional, fp32 [batch][M]): dbias += alpha * column sums of dZ - the bias gradient comes out of the same
 * launch (the kernel sums the dZ tiles it stages anyway), so no separate pass over dZ is needed for it.
 * Autograd backward of every nn.Linear / ParallelLinear on the path. */
size_t smx_linear_wgrad_workspace(int rows, int M, int K, int batch);
int smx_linear_wgrad(int dtype, const void* dZ, int64_t lddz, int64_t strideZ, const void* X, int64_t ldx,
                     int64_t strideX, float* dW, int64_t lddw, int64_t strideW, float* dbias, int rows, int M, int K,
                     int batch, float alpha, void* workspace, void* stream);

/* Deferred variant: only the split-K GEMM; the slabs (and, with want_bias, the bias partials behind them) stay in
 * `workspace` (same size query) for a later smx_reduce_jobs.  Outputs: *nslabs, *slab_stride (floats between slabs of
 * the [batch][M][K] weight image), *bias_offset (floats from the workspace start to the [nslabs][batch][M] bias
 * partials).  Returns SMX_EUNSUPPORTED for shapes the slab path cannot take (K % 4 != 0): use smx_linear_wgrad. */
int smx_linear_wgrad_partial(int dtype, const void* dZ, int64_t lddz, int64_t strideZ, const void* X, int64_t ldx,
                             int64_t strideX, int rows, int M, int K, int batch, int want_bias, void* workspace,
                             int32_t* nslabs, int64_t* slab_stride, int64_t* bias_offset, void* stream);

/* ALL the weight gradients of one encoder layer in one launch (bf16): for every item w
 *   slabs_w[s] (M_w x K_w, fp32) = dZ_w[rows of slice s]^T X_w[rows of slice s],     s < splits
 * and, with want_bias, bias partials [splits][M_w] (column sums of dZ_w) behind the slabs in the item's workspace
 * (smx_wgrad_group_workspace(M, K, splits) bytes, 16-byte aligned); fold them with smx_reduce_jobs
 * (src = workspace, src_stride = M*K, nsrc = splits; bias: src = workspace + splits*M*K floats, src_stride = M).
 * Every item reduces over the SAME `rows` frames (>= 64; the rows % 64 tail is staged zero-filled by the last split), M and K are
 * multiples of 256, operands 16-byte aligned with leading dimensions % 8 == 0.  One 512-thread workgroup per CU owns a
 * (weight, 256 x 256 tile, K slice) item; smx_wgrad_group_splits picks the slice count that fills the chip once.
 * Autograd backward (dW, db) of all the nn.Linear / Conv1d(k=1) modules of a ConformerEncoderLayer /
 * BranchformerEncoderLayer (Conformer.py:479-537, Branchformer.py:243-334) at once. */
#define SMX_WGRAD_GROUP_MAX 16
typedef struct smx_wgrad_item {
  const void* dZ; int64_t lddz;      /* (rows, M) gradient w.r.t. the layer's pre-activation                 */
  const void* X;  int64_t ldx;       /* (rows, K) the layer's input                                          */
  void* workspace;                   /* [splits][M*K] fp32 slabs, then [splits][M] bias partials             */
  int32_t M, K, want_bias, pad;
} smx_wgrad_item;
int smx_wgrad_group_splits(int rows, const smx_wgrad_item* items, int nitems);
size_t smx_wgrad_group_workspace(int M, int K, int splits);
int smx_wgrad_group(int dtype, int rows, const smx_wgrad_item* items, int nitems, int splits, void* stream);

/* Small batches (round 6): the same weight gradients WITHOUT split-K slabs - every 128 x 128 tile of every weight walks all `rows`
 * frames and adds its product INTO the gradient, dW[i] (M x K, float32, row stride lddw) += dZ_i^T X_i, dbias[i] (M) += column sums of
 * dZ_i (NULL: none).  No workspace, no smx_reduce_jobs pass; each gradient element has exactly one writer (bit-reproducible).  M, K
 * multiples of 128.  The slab form above is the one for long batches (its K-slices fill the chip; this one would not). */
typedef struct smx_wgrad_direct_item {
  const void* dZ; int64_t lddz;
  const void* X;  int64_t ldx;
  float* dW; int64_t lddw;
  float* dbias;
  int32_t M, K;
} smx_wgrad_direct_item;
int smx_wgrad_group_direct_ok(int rows, int M, int K);
int smx_wgrad_group_direct(int dtype, int rows, const smx_wgrad_direct_item* items, int nitems, void* stream);

/* One launch for many small fixed-order reductions
 *   dst[i*ldd + j] += alpha * sum_{s < nsrc} src[s*src_stride + i*src_ld + j]      (i < rows, j < cols; src_ld 0 = cols)
 * : weight-gradient slabs, bias partials, LayerNorm dgamma/dbeta partial rows.  `vec` = 1
 * when cols, ldd, src_ld, src_stride are multiples of 4 and src / dst are 16-byte aligned.  jobs and block_starts (njobs + 1 entries,
 * prefix sums of smx_reduce_job_blocks) live in device memory; tables can be cached while the pointers stay valid. */
typedef struct smx_reduce_job {
  const float* src; float* dst; int64_t src_stride; int64_t ldd; int32_t nsrc; int32_t rows; int32_t cols; float alpha;
  int32_t vec; int32_t src_ld;
} smx_reduce_job;
int smx_reduce_job_blocks(const smx_reduce_job* job_host);
int smx_reduce_jobs(const smx_reduce_job* jobs_dev, const int32_t* block_starts_dev, int njobs, int total_blocks,
                    void* stream);

/* Y = R + alpha * act(X W^T + b [+C0]) * mask ; thin wrapper over smx_gemm(NT).
 * Replaces summary_mixing.py:257 (global_proj * mask), :207/:210 (local/summary proj * mask),
 * :282-284 (merge with the per-utterance summary folded in as C0 = sbar W_s^T + b, SMX_C0_GROUP, div=T). */
int smx_linear_act_mask_fwd(int dtype, const void* X, int64_t ldx, const void* W, int64_t ldw, void* Y,
                            int64_t ldy, int N, int M, int K, const smx_epilogue* epi, void* stream);

/* dZ = dropout_mask(seed) * dY * mask * act'(Z)  (elementwise; the mask of the forward's fused dropout is regenerated
 * from the seed), plus fused parameter-gradient side reductions:
 *   dbias[m]          += sum_n dZ[n,m]                      (optional; per-strip partials in `workspace`,
 *                                                            smx_act_mask_bwd_workspace bytes, fixed-order reduce)
 *   dgroup[n/div, m]  += sum over the rows of group n/div   (fp32 atomics, optional; the backward of a
 *                                                            SMX_C0_GROUP side input)
 * Backward of the epilogue above (autograd of summary_mixing.py:207-284). dY is pre-scaled by alpha. */
size_t smx_act_mask_bwd_workspace(int N, int M);
int smx_act_mask_bwd(int dtype, const void* dY, int64_t lddy, const void* Z, int64_t ldz,
                     const uint8_t* row_mask, void* dZ, int64_t lddz, int N, int M, int act, float alpha,
                     float* dbias, float* dgroup, int64_t lddgroup, int group_div, float drop_p, uint64_t drop_seed, const uint64_t* epoch,
                     void* workspace, void* stream);

/* Masked mean over time: out[b,:] = sum_t S[b,t,:] * mask[b,t] / sum_t mask[b,t]   (fp32 out (B,D)).
 * S is (B*T, D) with leading dimension lds.  mask NULL => all valid.  scale_by_count=0 gives the plain sum.
 * inv_count (optional, (B) fp32) receives 1/sum_t mask[b,t].  `workspace` holds
 * smx_masked_mean_workspace(B,T,D) bytes (split-T partial sums, combined in a fixed order => bit-reproducible).
 * A row with zero valid frames yields NaN like the reference.  Replaces summary_mixing.py:218-220, :264-266,
 * :305-307.  This is the HBM-roofline kernel of BASELINE config 5. */
size_t smx_masked_mean_workspace(int B, int T, int D);
int smx_masked_mean_fwd(int dtype, const void* S, int64_t lds, const uint8_t* mask, float* out, float* inv_count,
                        int B, int T, int D, int scale_by_count, void* workspace, void* stream);
/* Backward / broadcast: dS[b,t,:] = g[b,:] * (inv_count ? inv_count[b] : 1) for every t (the row mask is
 * applied by the producer's smx_act_mask_bwd).  Also the forward `repeat` (summary_mixing.py:222,267), there with the
 * training dropout of the concatenated merge input fused in (drop_p > 0: mask = f(seed, row * D + col), the
 * smx_dropout indexing; summary_mixing.py:237-239). */
int smx_masked_mean_bwd(int dtype, const float* g, const float* inv_count, void* dS, int64_t ldds, int B, int T,
                        int D, float drop_p, uint64_t drop_seed, const uint64_t* epoch, void* stream);

/* Small batches (round 6): the masked mean over time AND its broadcast in ONE launch, fixed summation order (no atomics), for
 * smx_pool_bcast_ok(B, T, D) shapes (B * T <= 16 384 frames, T <= 4096):
 *   sum[b,:] = sum_t S[b,t,:] * mask_in[b,t];  mean_out[b,:] (optional) = sum (* 1 / count[b] when scale_by_count);
 *   inv_out[b] (optional) = 1 / count[b];
 *   dS[b,t,:] (optional) = D( value[b,:] * (inv_in ? inv_in[b] : 1) ) [* act'(Z[b,t,:]) * mask_out[b,t] when Z / mask_out are given]
 * i.e. smx_masked_mean_fwd followed by smx_masked_mean_bwd (forward `repeat` + the merge input's dropout, summary_mixing.py:218-222,
 * 237-239,264-267) or by smx_masked_mean_bwd_act (the backward of the same lines), without the workspace round trip and two of
 * the three launches.  Dropout (drop_p > 0) and the act / mask backward exclude each other. */
int smx_pool_bcast_ok(int B, int T, int D);
int smx_pool_bcast(int dtype, const void* S, int64_t lds, const uint8_t* mask_in, float* mean_out, const float* inv_in, float* inv_out,
                   void* dS, int64_t ldds, int B, int T, int D, int scale_by_count, float drop_p, uint64_t drop_seed,
                   const uint64_t* epoch, const void* Z, int64_t ldz, const uint8_t* mask_out, int act, void* stream);

/* Same broadcast with the activation / mask backward of the projection that produced the summary columns fused in:
 *   dS[b,t,:] = g[b,:] * inv_count[b] * act'(Z[b,t,:]) * row_mask[b,t]      (Z and/or row_mask given)
 * i.e. the dZ of `s = act(x W_s^T + b) * mask` (summary_mixing.py:210,257) without writing and re-reading the broadcast. */
int smx_masked_mean_bwd_act(int dtype, const float* g, const float* inv_count, void* dS, int64_t ldds, const void* Z,
                            int64_t ldz, const uint8_t* row_mask, int act, int B, int T, int D, void* stream);

/* DynChunk summary (sum_mask path, summary_mixing.py:224-235, :269-280) in O(T): frame t of chunk
 * c = t / chunk sees frames [max(0,(c-left)*chunk), min(T,(c+1)*chunk)) (left < 0: unlimited);
 * out[b,t,:] = sum over that window of S[b,.,:] / window length  (the denominator ignores padding, as the
 * reference's rowsum(sum_mask) does).  bwd is the transposed operator.  workspace:
 * smx_chunk_mean_workspace bytes (per-chunk fp32 sums). */
size_t smx_chunk_mean_workspace(int B, int T, int D, int chunk);
int smx_chunk_mean_fwd(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D,
                       int chunk, int left, void* workspace, void* stream);
int smx_chunk_mean_bwd(int dtype, const void* dOut, int64_t ldo, void* dS, int64_t lds, int B, int T, int D,
                       int chunk, int left, void* workspace, void* stream);

/* SummaryMixing-expdecay summary in O(T) (the reference materialises the (T,T) Laplace matrix gamma^|i-j|,
 * summary_mixing.py:316-365, and computes (M s) / rowsum(M), :233-235 - O(T^2)):
 *   fwd: out[b,t,:] = sum_j decay^|t-j| S[b,j,:] / sum_j decay^|t-j|   (j over ALL T frames: padding is ignored in the
 *        denominator exactly like the reference; padded frames of S are zero because the projection is masked);
 *   bwd: dS = M (dOut / rowsum(M))  (M is symmetric).  Two-sided exponential recurrences, chunked scan over T.
 * Only for sum_mask == None; with a DynChunk mask the reference's dense path is kept. */
size_t smx_expdecay_mean_workspace(int B, int T, int D);
int smx_expdecay_mean_fwd(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D, float decay,
                          void* workspace, void* stream);
int smx_expdecay_mean_bwd(int dtype, const void* dOut, int64_t ldo, void* dS, int64_t lds, int B, int T, int D, float decay,
                          void* workspace, void* stream);
/* (smx_layernorm_bwd with dgamma == dbeta == NULL leaves its partial rows [smx_layernorm_bwd_blocks(N)][2][D] in the
 *  workspace for smx_reduce_jobs: two jobs, src = ws (dgamma) and ws + D (dbeta), src_stride 2*D, rows 1, cols D.) */
int smx_layernorm_bwd_blocks(int N);
/* LayerNorm over the last dim with an optional fused activation: Y = act(LN(X))
 * (torch.nn.LayerNorm; Conformer.py:146,152-153,475-476,738).  stats (N,2) fp32 = (mean, rstd), optional in fwd.
 * bwd: dX = R + LNbwd(dY * act'(LN(X))) (R optional residual-gradient, dtype T; LN(X) is recomputed from the
 * stats); dgamma/dbeta += per-block partial sums (in `workspace`, smx_layernorm_bwd_workspace bytes) reduced in a
 * fixed order: bit-reproducible, no atomics. */
int smx_layernorm_fwd(int dtype, const void* X, int64_t ldx, const float* gamma, const float* beta, void* Y,
                      int64_t ldy, float* stats, int N, int D, float eps, int act, void* stream);
/* The same with a float32 input X and an output of dtype `dtype` (fp32 residual stream -> bf16 GEMM input), and the
 * backward with a float32 X (dY, R, dX, dX2 dtype `dtype`): torch.nn.LayerNorm under autocast. */
int smx_layernorm_fwd_x32(int dtype, const float* X, int64_t ldx, const float* gamma, const float* beta, void* Y,
                          int64_t ldy, float* stats, int N, int D, float eps, int act, void* stream);
/* Two LayerNorms in ONE pass over the float32 residual stream: Y1 = LN1(X) (float32: the layer-final norm2, Conformer.py:536 -
 * the next layer's stream input) and Y2 = LN2(Y1) (dtype2: the LayerNorm of the next layer's first feed-forward module,
 * Conformer.py:458-459,507); stats1 / stats2 (N,2) = (mean, rstd), optional.  Equal to smx_layernorm_fwd (fp32) followed by
 * smx_layernorm_fwd_x32 to an ulp; Y1 is not re-read and is stored with the non-temporal hint.  D % 4 == 0, D <= 2048, 16-byte aligned rows (SMX_EUNSUPPORTED otherwise). */
int smx_layernorm_fwd_pair_x32(int dtype2, const float* X, int64_t ldx, const float* gamma1, const float* beta1, float eps1,
                               float* Y1, int64_t ldy1, float* stats1, const float* gamma2, const float* beta2, float eps2,
                               void* Y2, int64_t ldy2, float* stats2, int N, int D, void* stream);
size_t smx_layernorm_bwd_workspace(int N, int D);
int smx_layernorm_bwd(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                      const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX,
                      int64_t lddx, float* dgamma, float* dbeta, int N, int D, void* workspace, void* stream);
/* The same with a second output written from the registers that hold dX:
 *   dX2 = alpha2 * Dropout(dX; drop_p2, drop_seed2, index n*D + c) * row_mask2[n]        (D <= 2048)
 * - the first thing the next backward block does to this gradient (Conformer.py:507,536: the FFN module's 1/2 * dropout;
 * :146-152,327-331: the conv module's dropout and padding mask), so that block needs no elementwise pass of its own. */
int smx_layernorm_bwd2(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                       const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX,
                       int64_t lddx, float* dgamma, float* dbeta, int N, int D, void* workspace, void* dX2, int64_t lddx2,
                       float alpha2, const uint8_t* row_mask2, float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream);
int smx_layernorm_bwd2_x32(int dtype, const void* dY, int64_t lddy, const float* X, int64_t ldx, const float* gamma,
                           const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX,
                           int64_t lddx, float* dgamma, float* dbeta, int N, int D, void* workspace, void* dX2, int64_t lddx2,
                           float alpha2, const uint8_t* row_mask2, float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream);
/* The same with the incoming gradient given as `nslab` float32 split-K slabs ((N, D) each, slab_stride elements apart; added in slab
 * order) - the dgrad of the Linear behind the LayerNorm computed by smx_gemm_panel_slabs: reducer and LayerNorm backward in one launch.
 * x_f32: X is the float32 residual stream.  dgamma / dbeta partial rows stay in `workspace` (smx_reduce_jobs). */
int smx_layernorm_bwd2_slabs(int dtype, const float* slabs, int nslab, int64_t slab_stride, const void* X, int64_t ldx, int x_f32,
                             const float* gamma, const float* beta, int act, const float* stats, const void* R, int64_t ldr, void* dX,
                             int64_t lddx, int N, int D, void* workspace, void* dX2, int64_t lddx2, float alpha2, const uint8_t* row_mask2,
                             float drop_p2, uint64_t drop_seed2, const uint64_t* epoch, void* stream);
/* LayerNorm backward THROUGH the activation that produced the LayerNorm's input: X = zact(Z) (Z the saved pre-activation),
 *   dZ = zact'(Z) * LNbwd(dY)                      (act must be SMX_ACT_NONE: a LayerNorm without a fused activation of its own)
 * - the CSGU of the Branchformer's cgMLP normalises the gate half of GELU(channel_proj1(x)) (Branchformer.py:84-96 via the
 * upstream ConvolutionalSpatialGatingUnit), so the gradient of that half reaches channel_proj1's dZ without the separate
 * activation-backward pass over it.  bf16, D <= 2048, D % 8 == 0, 16-byte aligned rows (else SMX_EUNSUPPORTED: run
 * smx_layernorm_bwd + smx_act_mask_bwd).  dgamma / dbeta as in smx_layernorm_bwd (NULL: partial rows stay in `workspace`). */
int smx_layernorm_bwd_preact(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, const float* gamma,
                             const float* beta, int act, const float* stats, const void* Z, int64_t ldz, int zact,
                             void* dX, int64_t lddx, float* dgamma, float* dbeta, int N, int D, void* workspace, void* stream);

/* Fused GLU + depthwise Conv1d over time (Conformer.py:131-145,317-325):
 *   u[b,t,c] = P[b,t,c] * sigmoid(P[b,t,D+c]);  Y[b,t,c] = bias[c] + sum_j w[c,j] u[b,t+j-(k-1)/2,c]
 * glu=0: u = P (ldp >= D).  pad SMX_PAD_ZERO (Conformer) | SMX_PAD_REFLECT (Branchformer CSGU).
 * chunk > 0: Dynamic Chunk Convolution (Conformer.py:190-313): inputs at or beyond the end of the output
 * frame's own chunk read as zero.  gate != NULL: Y *= gate (CSGU x1*x2).  w (D,k) fp32, bias (D) fp32. */
int smx_dwconv1d_glu_fwd(int dtype, const void* P, int64_t ldp, const float* w, const float* bias,
                         const void* gate, int64_t ldg, void* Y, int64_t ldy, int B, int T, int D, int k, int glu,
                         int pad_mode, int chunk, void* stream);
/* Same with an inverted dropout of the output fused in (mask index = global row * D + channel, as smx_dropout on Y):
 * the CSGU's own dropout (upstream ConvolutionalSpatialGatingUnit.forward).  Only the rolling CSGU kernel carries it
 * (bf16, k = 31, gate, reflect padding, D % 64 == 0, aligned rows); SMX_EUNSUPPORTED otherwise - run smx_dropout. */
int smx_dwconv1d_glu_fwd_drop(int dtype, const void* P, int64_t ldp, const float* w, const float* bias,
                              const void* gate, int64_t ldg, void* Y, int64_t ldy, int B, int T, int D, int k, int glu,
                              int pad_mode, int chunk, float drop_p, uint64_t drop_seed, const uint64_t* epoch, void* stream);
/* dP (same shape as P), dw/dbias += ; dgate optional (= dY * conv).  workspace: smx_dwconv1d_glu_bwd_workspace bytes
 * (per-block partial tap gradients of the fast path, reduced in a fixed order; NULL selects the generic kernel).
 * dw == NULL (and dbias == NULL): the partial rows [smx_dwconv1d_glu_bwd_partial_rows][D][k+1] (taps, then the bias term)
 * stay in the workspace for smx_reduce_jobs (two jobs with src_ld = k+1); SMX_EUNSUPPORTED outside the k = 31 path. */
size_t smx_dwconv1d_glu_bwd_workspace(int B, int T, int D, int k);
/* how many partial rows the call below leaves in the workspace for these arguments (the nsrc of the two reduction jobs) */
int smx_dwconv1d_glu_bwd_partial_rows(int dtype, int B, int T, int D, int k, int glu, int pad_mode, int chunk, int has_gate);
int smx_dwconv1d_glu_bwd(int dtype, const void* dY, int64_t lddy, const void* P, int64_t ldp, const float* w,
                         const float* bias, const void* gate, int64_t ldg, void* dP, int64_t lddp, void* dgate,
                         int64_t lddg, float* dw, float* dbias, int B, int T, int D, int k, int glu, int pad_mode,
                         int chunk, void* workspace, void* stream);

/* ---- front-end (arithmetic lives in un-vendored SpeechBrain: parity unpinned, spec = oracle) ----
 * Log-mel filterbank (speechbrain.lobes.features.Fbank, ...transducer.yaml:171-175):
 *   smx_frame_window : frames[b*T+t, j] = window[j] * wav[b, t*hop + j - n_fft/2]  (center=True, zero pad), fp32
 *   DFT              : smx_gemm(NT, F32) of the frames with a [cos | -sin] basis (exact-fp32 MFMA)
 *   smx_mel_db       : power spectrum -> mel filterbank (n_mels x n_bins fp32) -> 10 log10(max(., amin)) -> per-utterance
 *                      top_db clamp; spec (B*T, lds) holds re at column f and im at column im_off + f.
 * Conv subsampling (ConvolutionFrontEnd, ...yaml:247-254): 3x3 / stride 2 / reflect pad 1, channels-last:
 *   smx_im2col_s2 : x (B,T,F,C) -> col (B*ceil(T/2)*ceil(F/2), Kp), column (dt*3+df)*C + c, zero beyond 9*C
 *   conv          : smx_gemm(NT) col x W^T + bias, then smx_layernorm_fwd over (F2*Cout) with fused LeakyReLU
 *   smx_col2im_s2 : backward of im2col (gather form, no atomics). */
/* The same DFT without the frame matrix and at half the reduction length: frames are rows of the zero-padded waveform
 * (wav_padded (B, ldw): n_fft / 2 zeros in front, frame t of utterance b = samples [t hop, t hop + n_fft)), the (symmetric) window is
 * folded into the bases, and the real DFT is split into its cosine part on s[j] = x[j] + x[n - j] (basis_cos (rows_basis, n/2 + 4):
 * window[j] cos(2 pi k j / n) for j <= n/2, zero behind) and its sine part on d[j] = x[j] - x[n - j] (basis_sin (rows_basis, n/2):
 * -window[j] sin(2 pi k j / n)); s and d are formed inside the GEMM's operand loader.  spec (B*T, lds): re at column k, im at im_off + k. */
int smx_dft_frames(const float* wav_padded, int64_t ldw, const float* basis_cos, const float* basis_sin, float* spec, int64_t lds,
                   int im_off, int B, int T, int n_fft, int hop, int rows_basis, void* stream);
int smx_frame_window(const float* wav, int64_t ldw, const float* window, float* frames, int B, int L, int T, int n_fft,
                     int hop, void* stream);
size_t smx_fbank_workspace(int B, int T, int n_mels);
int smx_mel_db(int out_dtype, const float* spec, int64_t lds, int im_off, const float* fb, int n_bins, int n_mels,
               float amin, float top_db, void* out, int B, int T, void* workspace, void* stream);
int smx_im2col_s2(int dtype, const void* x, void* col, int B, int T, int F, int C, int Kp, void* stream);
int smx_col2im_s2(int dtype, const void* dcol, void* dx, int B, int T, int F, int C, int Kp, void* stream);
/* The first conv block in one pass per direction (1 input channel, O = 64, F a multiple of 16 up to 160; SMX_EUNSUPPORTED
 * otherwise: im2col + smx_linear_k16_fwd / smx_gemm + smx_layernorm_*).  X (B,T,F) and Y / dA (B*ceil(T/2), (F/2)*O) dtype T;
 * W9 (O, 9) fp32 taps (dt*3 + df), gamma / beta ((F/2)*O) fp32.
 *   fwd: Y = act(LayerNorm_row(conv3x3_s2_reflect(X) + bias) * gamma + beta), stats (rows, 2) = mean | rstd
 *   bwd: grads[(F/2)*O | (F/2)*O | O*9 | O] += dgamma | dbeta | dW9 | dbias, from dA and X alone (the convolution is recomputed;
 *        nothing of activation size is written); `workspace` = smx_conv1_ln_workspace bytes of per-workgroup partial rows,
 *        folded in a fixed order. */
size_t smx_conv1_ln_workspace(int B, int T, int F, int O);
int smx_conv1_ln_fwd(int dtype, const void* X, const float* W9, const float* bias, const float* gamma, const float* beta, float eps,
                     int act, void* Y, float* stats, int B, int T, int F, int O, void* stream);
int smx_conv1_ln_bwd(int dtype, const void* dA, const void* X, const float* W9, const float* bias, const float* gamma,
                     const float* beta, const float* stats, int act, float* grads, void* workspace, int B, int T, int F, int O,
                     void* stream);
/* Y (N, M) = X (N, 16) W (M, 16)^T + bias: the first block's convolution (9 taps of one input channel in 16 columns) on
 * the VALU, bf16 in / out, fp32 accumulation; dense rows.  SMX_EUNSUPPORTED for other shapes (use smx_gemm). */
int smx_linear_k16_fwd(int dtype, const void* X, const void* W, const float* bias, void* Y, int64_t N, int M, void* stream);
/* The second block's convolution and its weight gradient WITHOUT the (rows, 9 C) patch matrix: the MFMA GEMM kernels gather
 * their operand from the channels-last input X (B,T,F,C) themselves (one tap per K tile / column tile).  Built for bf16, C = 64
 * (SMX_EUNSUPPORTED otherwise: smx_im2col_s2 + smx_gemm / smx_linear_wgrad).  Wg / dWg (O, Kp) GEMM layout (column
 * (dt*3+df)*C + c); Y / dY (B*ceil(T/2)*ceil(F/2), O) dense.
 *   fwd  : Y = conv(X) + bias
 *   wgrad: dWg (fp32) += dY^T patches(X), dbias += column sums of dY; slab split-K + fixed-order reduction (`workspace` =
 *          smx_conv2d_s2_wgrad_workspace bytes, 16-byte aligned). */
int smx_conv2d_s2_fwd(int dtype, const void* X, const void* Wg, const float* bias, void* Y, int B, int T, int F, int C, int O,
                      int Kp, void* stream);
size_t smx_conv2d_s2_wgrad_workspace(int B, int T, int F, int C, int O);
int smx_conv2d_s2_wgrad(int dtype, const void* dY, const void* X, float* dWg, float* dbias, int B, int T, int F, int C, int O,
                        int Kp, void* workspace, void* stream);
/* Direct input gradient of that convolution, without the (rows, 9 C) column matrix: dX (B,T,F,C) from dY
 * (B,ceil(T/2),ceil(F/2),O) and the GEMM-layout weight Wg (O, Kp) (column (dt*3+df)*C + c).  Built for the recipe's second
 * block (bf16, C = 64, O = 32; SMX_EUNSUPPORTED otherwise: use the dgrad GEMM + smx_col2im_s2).  MFMA, no atomics. */
int smx_conv2d_s2_dgrad(int dtype, const void* dY, const void* Wg, void* dX, int B, int T, int F, int C, int O, int Kp,
                        void* stream);

/* y = a*x (+ b*y0): generic strided elementwise helper (dtype T). */
int smx_axpby(int dtype, float a, const void* X, int64_t ldx, float b, const void* Y0, int64_t ldy0, void* Y,
              int64_t ldy, int N, int D, void* stream);
/* Inverted dropout with a counter-based generator: Y[n,c] = keep(seed, n*D+c) ? X[n,c]/(1-p) : 0 (in place allowed).
 * The mask is a pure function of (seed, element index): calling it again on the gradient with the same seed is the
 * backward.  Replaces nn.Dropout at summary_mixing.py:238,283, Conformer.py:157,461-472, TransformerASR.py:353. */
int smx_dropout(int dtype, const void* X, int64_t ldx, void* Y, int64_t ldy, int N, int D, float p, uint64_t seed, const uint64_t* epoch,
                void* stream);
/* Y[n,:] += table[n % R,:] (fp32 table): abs-sine positional encoding added after the input dropout
 * (TransformerASR.py:547-549). */
int smx_add_rowtable(int dtype, void* Y, int64_t ldy, const float* table, int R, int N, int D, void* stream);
/* fp32 -> T cast of a flat buffer (bf16 shadow weights). */
int smx_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream);
int smx_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream);

/* Fused AdamW over a flat fp32 parameter buffer (decoupled weight decay, torch.optim.AdamW semantics;
 * recipes/LibriSpeech/.../conformer_summarymixing_transducer.yaml:395-399).  grad_scale multiplies the
 * gradient first (1/world_size and the global-norm clip factor); `gscale_dev` (optional, device fp32[1])
 * is multiplied in as well so the clip factor never visits the host.  shadow (optional) receives the
 * updated parameters cast to bf16. */
int smx_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, const float* gscale_dev, const uint64_t* step_dev, void* stream);
/* ---- InputNormalization between the filterbank and the CNN front-end (recipe key `normalize`:
 * speechbrain.processing.features.InputNormalization, …transducer.yaml:167-169; upstream-only arithmetic). ----
 * smx_utt_meanstd: mean[b,f] and unbiased std[b,f] over the frames t < len[b] (std clamped below by eps; 0 / 1 when the
 * respective normalisation is off).  smx_stats_combine: glob = (1-w)*glob + w*mean_b(cur) (w = 1 replaces: the first
 * batch / norm_type "batch").  smx_colnorm: Y = (X - mean[b*stat_stride + f]) / std[...] over (B, T, F) rows b*T + t
 * (stat_stride 0 = shared statistics: "global" / "batch"; F = per-utterance: "sentence"). */
int smx_utt_meanstd(int dtype, const void* X, int64_t ldx, const int32_t* len, float* mean, float* std, int B, int T, int F,
                    int mean_norm, int std_norm, float eps, void* stream);
int smx_stats_combine(const float* cur_mean, const float* cur_std, int B, int F, float* glob_mean, float* glob_std, float weight,
                      void* stream);
int smx_colnorm(int dtype, const void* X, int64_t ldx, const float* mean, const float* std, int64_t stat_stride, void* Y,
                int64_t ldy, int B, int T, int F, void* stream);

/* ---- CTC head after the encoder (recipe keys `log_softmax`, `ctc_cost`:
 * …/LibriSpeech/ASR/transducer/hparams/conformer_summarymixing_transducer.yaml:297-298,331). ----
 * Y = log_softmax(X) over the last dim (N rows of V); bwd: dX = dY - exp(Y) * rowsum(dY).
 * Replaces speechbrain.nnet.activations.Softmax(apply_log=True). */
int smx_log_softmax_fwd(int dtype, const void* X, int64_t ldx, void* Y, int64_t ldy, int N, int V, void* stream);
int smx_log_softmax_bwd(int dtype, const void* dY, int64_t lddy, const void* Y, int64_t ldy, void* dX, int64_t lddx, int N,
                        int V, void* stream);
/* CTC negative log-likelihood per utterance and its gradient; replaces torch.nn.functional.ctc_loss(...,
 * zero_infinity=True) as called by speechbrain.nnet.losses.ctc_loss.
 *   log_probs (B, T, V) row-major with leading dimension ldlp (rows b*T + t), dtype T (log-softmax outputs);
 *   targets int32 (B, Smax) padded, in_len / tgt_len int32 (B) absolute lengths, blank < V;
 *   fwd: nll[b] = -log p(targets_b | x_b)  (+inf when no alignment exists); keeps the forward variables in `workspace`
 *        (smx_ctc_workspace bytes) for the backward;
 *   bwd: grad[b,t,v] = gscale[b] * (exp(lp) - exp(log sum_{s: l'_s = v} alpha*beta/y + nll[b])), zero for t >= in_len[b]
 *        and for utterances with infinite nll (the convention of torch's ctc_loss backward: the gradient w.r.t. the
 *        unnormalised logits; pushing it through smx_log_softmax_bwd leaves it unchanged).  gscale[b] carries the
 *        reduction (1/B, 1/tgt_len, ...) times the upstream gradient.  Must follow smx_ctc_loss_fwd on the same workspace.
 *        The per-label sums follow each label's occurrence chain in a fixed order: no atomics, bit-reproducible. */
size_t smx_ctc_workspace(int B, int T, int Smax);
int smx_ctc_loss_fwd(int dtype, const void* log_probs, int64_t ldlp, const int32_t* targets, const int32_t* in_len,
                     const int32_t* tgt_len, int B, int T, int V, int Smax, int blank, float* nll, void* workspace, void* stream);
int smx_ctc_loss_bwd(int dtype, const void* log_probs, int64_t ldlp, const int32_t* targets, const int32_t* in_len,
                     const int32_t* tgt_len, int B, int T, int V, int Smax, int blank, const float* nll, const float* gscale,
                     void* grad, int64_t ldg, void* workspace, void* stream);

/* ---- Split-K over WORKGROUPS for the long reductions of a small batch (round 6; the recipe's 10 x 375 frames) -------------------
 * smx_gemm_panel_slabs: slab[s] (N x M, float32) = A[:, s K : (s + 1) K] . W_s^T for s < nslice on the panel-resident kernel
 *   (A (N, nslice K) bf16; Wpacked = nslice consecutive smx_weight_pack images, image s = the weight's K-slice s, packed WITHOUT a bias;
 *   K = 256 or 512 per slice, M %% 64 == 0, M <= 512).  slabs: nslice * N * M floats, caller-owned.
 * smx_slab_epilogue: C = epilogue(sum_s slab[s]) with the slabs added in a fixed order (bit-reproducible, no atomics) and the
 *   smx_gemm epilogue fields bias, act (+ z saved), drop_*, alpha, row_mask, res (SMX_IO_RES_F32), out_mode (SMX_OUT_T / SMX_OUT_F32)
 *   and SMX_EPI_LN_FWD (lnf_*, SMX_IO_LNFY_F32, lnf2_*) - the LayerNorm(s) that follow the Linear (Conformer.py:458-476,507,530-536)
 *   run on the row in registers: one launch instead of GEMM epilogue + standalone LayerNorm.  Other fields: SMX_EUNSUPPORTED. */
int smx_gemm_panel_slabs_ok(int dtype, int N, int M, int K, int nslice);
int smx_gemm_panel_slabs(int dtype, const void* A, int64_t lda, const void* Wpacked, float* slabs, int N, int M, int K, int nslice,
                         void* stream);
int smx_slab_epilogue_ok(int dtype, int N, int M, int nslab);
int smx_slab_epilogue(int dtype, const float* slabs, int nslab, int64_t slab_stride, void* C, int64_t ldc, int N, int M,
                      const smx_epilogue* epi, void* stream);

/* ---- Sequence-parallel shards (summarymixing_amd/sequence_parallel.py): the boundary arithmetic of the two O(T) summaries inside the
 * kernels (round 6; it ran as float32 elementwise passes over the activations on the host side before).
 * smx_chunk_mean_sharded: the shard holds the whole chunks [c_off, c_off + T / chunk).  phase 1: chunk sums into `workspace`
 *   ((B, T / chunk, D) float32; reverse: scaled by 1 / the GLOBAL window length; left < 0: running sums) - rows of it are what the
 *   shards exchange.  phase 2: window combine + carry[b][c - carry_c0] for carry_c0 <= c < carry_c0 + carry_n (carry_n == 0: one
 *   (B, D) row for every chunk), forward divided by the GLOBAL window length.
 * smx_expdecay_mean_sharded: frames [t_off, t_off + T) of T_glob.  phase 1: ends (2, B, D) <- the states leaving the shard; phase 2:
 *   ends = the states ENTERING it (folded by the caller from the gathered ones), out = the operator with global denominators. */
int smx_chunk_mean_sharded(int dtype, const void* X, int64_t ldx, void* out, int64_t ldo, int B, int T, int D, int chunk, int left,
                           int reverse, int c_off, int phase, const float* carry, int carry_c0, int carry_n, void* workspace, void* stream);
int smx_expdecay_mean_sharded(int dtype, const void* S, int64_t lds, void* out, int64_t ldo, int B, int T, int D, float decay, int mode,
                              int t_off, int T_glob, int phase, float* ends, void* workspace, void* stream);

/* Device step counter (one uint64 in device memory) - an explicit ARGUMENT of every call that uses it, never library
 * state: `epoch` of smx_dropout / smx_masked_mean_bwd(_act) / smx_act_mask_bwd / smx_layernorm_bwd2 /
 * smx_dwconv1d_glu_fwd_drop, smx_epilogue.epoch of the GEMMs, `step_dev` of smx_adamw_step.  With a non-NULL counter a
 * fused / standalone dropout mixes the counter's current value into its seed and smx_adamw_step with step <= 0 takes its
 * bias-correction step from it: a whole training step can then be captured ONCE in a hipGraph (all kernel arguments
 * constant) and replayed - masks and bias correction still advance, because the counter does (smx_step_counter_add is
 * itself a captured kernel).  NULL = the plain by-value behaviour.  The library holds no process-global state besides
 * the read-once smx_config and the per-thread error string. */
int smx_step_counter_add(uint64_t* dev_counter, uint64_t inc, void* stream);
/* id[0] = the id of the hipGraph capture `stream` is recording into (hipStreamGetCaptureInfo), 0 when it is not capturing.  Host
 * bookkeeping only (no launch): the host side stamps cached packed weight images (smx_weight_pack) with it, so that EVERY capture
 * contains its own pack launches and the first eager call after a capture re-packs (round-5 advisor finding). */
int smx_stream_capture_id(void* stream, uint64_t* id);
/* out[0] += sum(x^2) — global grad-norm for clipping (zero out[0] first).  Fixed summation order (per-block partials in
 * `workspace`, smx_sumsq_workspace() bytes, folded by one block; no atomics): data-parallel ranks holding the same
 * all-reduced gradients get bit-identical norms, clip factors and weights. */
size_t smx_sumsq_workspace(void);
int smx_sumsq(const float* x, int64_t n, float* out, void* workspace, void* stream);
/* out[0] = min(1, max_norm / (sqrt(sumsq[0]) * inv_scale + 1e-6)) : clip factor computed on device.  `out` is fp32[2]:
 * when sumsq[0] is NaN / Inf (a bad batch: bf16 overflow, a zero-length utterance) out[0] = 0 and out[1] += 1 (skipped-step
 * counter); smx_adamw_step treats a device factor of exactly 0 as "skip the update" (no decay, no moments, no shadow
 * refresh) - the behaviour of SpeechBrain's Brain.check_gradients for non-finite gradients. */
int smx_clip_factor(const float* sumsq, float max_norm, float inv_scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMX_H_ */
