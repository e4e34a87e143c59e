"""ctypes binding of libsmx.so (include/smx.h).  The product path has NO fallback: if the HIP library is
missing or a call fails, a RuntimeError is raised."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMX_LIB: another build of the same library (e.g. libsmx_diag.so = csrc/build.sh with SMX_DIAG=1, for the ablation tools)
LIB_PATH = os.environ.get("SMX_LIB") or os.path.join(_HERE, "libsmx.so")

c_i, c_i64, c_f, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# enums of include/smx.h
F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SWISH, ACT_LEAKY_RELU, ACT_RELU = 0, 1, 2, 3, 4
C0_NONE, C0_ROW, C0_GROUP, C0_MOD = 0, 1, 2, 3
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
OUT_T, OUT_F32, OUT_ATOMIC_F32 = 0, 1, 2
EPI_C0_POST = 1
EPI_ACT_GRAD = 2
EPI_LN_BWD = 4
EPI_LN_FWD = 8
IO_RES_F32, IO_LNX_F32, IO_LNFY_F32 = 1, 2, 4
PAD_ZERO, PAD_REFLECT = 0, 1
ACTS = {"none": ACT_NONE, "identity": ACT_NONE, "gelu": ACT_GELU, "swish": ACT_SWISH,
        "leaky_relu": ACT_LEAKY_RELU, "relu": ACT_RELU}


class ReduceJob(ctypes.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("src_stride", c_i64), ("ldd", c_i64), ("nsrc", ctypes.c_int32),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("alpha", c_f), ("vec", ctypes.c_int32),
                ("src_ld", ctypes.c_int32)]


class WgradItem(ctypes.Structure):
    _fields_ = [("dZ", c_vp), ("lddz", c_i64), ("X", c_vp), ("ldx", c_i64), ("workspace", c_vp),
                ("M", ctypes.c_int32), ("K", ctypes.c_int32), ("want_bias", ctypes.c_int32), ("pad", ctypes.c_int32)]


class WgradDirectItem(ctypes.Structure):
    """smx_wgrad_direct_item of include/smx.h."""
    _fields_ = [("dZ", c_vp), ("lddz", c_i64), ("X", c_vp), ("ldx", c_i64), ("dW", c_vp), ("lddw", c_i64), ("dbias", c_vp),
                ("M", ctypes.c_int32), ("K", ctypes.c_int32)]


WGRAD_GROUP_MAX = 16


class GemmPlan(ctypes.Structure):
    """smx_gemm_plan of include/smx.h: the instantiation smx_gemm would launch (smx_gemm_plan_query)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("kernel", "a_kc", "b_kc", "tile_n", "tile_m", "vec", "lnf", "gather")]


class Epilogue(ctypes.Structure):
    _fields_ = [("bias", c_vp), ("bias_batch_stride", c_i64),
                ("c0", c_vp), ("ldc0", c_i64), ("c0_mode", ctypes.c_int32), ("c0_div", ctypes.c_int32),
                ("act", ctypes.c_int32), ("out_mode", ctypes.c_int32),
                ("z", c_vp), ("ldz", c_i64),
                ("row_mask", c_vp),
                ("res", c_vp), ("ldr", c_i64),
                ("alpha", c_f), ("flags", ctypes.c_int32),
                ("drop_p", c_f), ("drop_cols", ctypes.c_int32), ("drop_seed", ctypes.c_uint64),
                ("colsum", c_vp), ("workspace", c_vp),
                ("ln_x", c_vp), ("ln_ldx", c_i64), ("ln_stats", c_vp), ("ln_gamma", c_vp), ("ln_partial", c_vp),
                ("ln_dx2", c_vp), ("ln_lddx2", c_i64), ("ln_mask2", c_vp), ("ln_alpha2", c_f), ("ln_drop_p2", c_f),
                ("ln_drop_seed2", ctypes.c_uint64),
                ("lnf_gamma", c_vp), ("lnf_beta", c_vp), ("lnf_y", c_vp), ("lnf_ldy", c_i64), ("lnf_stats", c_vp),
                ("lnf_eps", c_f), ("lnf_act", ctypes.c_int32),
                ("io_flags", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("epoch", c_vp),
                ("lnf2_gamma", c_vp), ("lnf2_beta", c_vp), ("lnf2_y", c_vp), ("lnf2_ldy", c_i64), ("lnf2_stats", c_vp),
                ("lnf2_eps", c_f), ("pad2_", ctypes.c_int32)]


class PackJob(ctypes.Structure):
    """smx_pack_job of include/smx.h (one entry of smx_weight_pack_jobs' device table)."""
    _fields_ = [("W", c_vp), ("ldw", c_i64), ("bias", c_vp), ("packed", c_vp), ("M", ctypes.c_int32), ("K", ctypes.c_int32),
                ("transposed", ctypes.c_int32), ("block_start", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/smx.h one to one (tests/test_abi.py checks the export list)
SIGNATURES = {
    "smx_version": (c_i, []),
    "smx_last_error": (ctypes.c_char_p, []),
    "smx_gemm": (c_i, [c_i, c_i, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i,
                       ctypes.POINTER(Epilogue), c_vp]),
    "smx_gemm_plan_query": (c_i, [c_i, c_i, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i,
                                  ctypes.POINTER(Epilogue), ctypes.POINTER(GemmPlan)]),
    "smx_gemm_colsum_workspace": (c_sz, [c_i, c_i]),
    "smx_gemm_panel_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "smx_weight_pack_bytes": (c_sz, [c_i, c_i]),
    "smx_weight_pack": (c_i, [c_i, c_vp, c_i64, c_i, c_vp, c_i, c_i, c_vp, c_vp]),
    "smx_weight_pack_job_blocks": (c_i, [c_i, c_i]),
    "smx_weight_pack_jobs": (c_i, [c_i, c_vp, c_i, c_i, c_vp]),
    "smx_gemm_panel": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_i64, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_vp]),
    "smx_gemm_ln_fused_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "smx_gemm_ln_pair_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "smx_linear_wgrad_workspace": (c_sz, [c_i, c_i, c_i, c_i]),
    "smx_linear_wgrad": (c_i, [c_i, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i, c_i, c_i, c_i,
                               c_f, c_vp, c_vp]),
    "smx_linear_wgrad_partial": (c_i, [c_i, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_vp,
                                       ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_vp]),
    "smx_wgrad_group_splits": (c_i, [c_i, ctypes.POINTER(WgradItem), c_i]),
    "smx_wgrad_group_workspace": (c_sz, [c_i, c_i, c_i]),
    "smx_wgrad_group": (c_i, [c_i, c_i, ctypes.POINTER(WgradItem), c_i, c_i, c_vp]),
    "smx_reduce_job_blocks": (c_i, [ctypes.POINTER(ReduceJob)]),
    "smx_reduce_jobs": (c_i, [c_vp, c_vp, c_i, c_i, c_vp]),
    "smx_layernorm_bwd_blocks": (c_i, [c_i]),
    "smx_linear_act_mask_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i,
                                      ctypes.POINTER(Epilogue), c_vp]),
    "smx_act_mask_bwd_workspace": (c_sz, [c_i, c_i]),
    "smx_act_mask_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i, c_i, c_i, c_f, c_vp, c_vp,
                               c_i64, c_i, c_f, ctypes.c_uint64, c_vp, c_vp, c_vp]),
    "smx_masked_mean_workspace": (c_sz, [c_i, c_i, c_i]),
    "smx_masked_mean_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "smx_masked_mean_bwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_i, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_pool_bcast_ok": (c_i, [c_i, c_i, c_i]),
    "smx_pool_bcast": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint64, c_vp, c_vp, c_i64,
                             c_vp, c_i, c_vp]),
    "smx_masked_mean_bwd_act": (c_i, [c_i, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "smx_chunk_mean_workspace": (c_sz, [c_i, c_i, c_i, c_i]),
    "smx_chunk_mean_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "smx_chunk_mean_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "smx_expdecay_mean_workspace": (c_sz, [c_i, c_i, c_i]),
    "smx_expdecay_mean_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_f, c_vp, c_vp]),
    "smx_expdecay_mean_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_f, c_vp, c_vp]),
    "smx_layernorm_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i, c_i, c_f, c_i, c_vp]),
    "smx_layernorm_fwd_x32": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i, c_i, c_f, c_i, c_vp]),
    "smx_layernorm_fwd_pair_x32": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_f, c_vp, c_i64, c_vp, c_vp, c_vp, c_f, c_vp, c_i64, c_vp, c_i, c_i, c_vp]),
    "smx_layernorm_bwd2_x32": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                     c_vp, c_i, c_i, c_vp, c_vp, c_i64, c_f, c_vp, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_layernorm_bwd_preact": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_i, c_vp, c_i64,
                                       c_vp, c_vp, c_i, c_i, c_vp, c_vp]),
    "smx_layernorm_bwd_workspace": (c_sz, [c_i, c_i]),
    "smx_layernorm_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                c_vp, c_i, c_i, c_vp, c_vp]),
    "smx_layernorm_bwd2": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                 c_vp, c_i, c_i, c_vp, c_vp, c_i64, c_f, c_vp, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_dwconv1d_glu_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_vp]),
    "smx_dwconv1d_glu_fwd_drop": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i,
                                        c_i, c_i, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_dwconv1d_glu_bwd_workspace": (c_sz, [c_i, c_i, c_i, c_i]),
    "smx_dwconv1d_glu_bwd_partial_rows": (c_i, [c_i] * 9),
    "smx_dwconv1d_glu_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                   c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "smx_dft_frames": (c_i, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_frame_window": (c_i, [c_vp, c_i64, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_fbank_workspace": (c_sz, [c_i, c_i, c_i]),
    "smx_mel_db": (c_i, [c_i, c_vp, c_i64, c_i, c_vp, c_i, c_i, c_f, c_f, c_vp, c_i, c_i, c_vp, c_vp]),
    "smx_im2col_s2": (c_i, [c_i, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_col2im_s2": (c_i, [c_i, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_conv1_ln_workspace": (c_sz, [c_i, c_i, c_i, c_i]),
    "smx_conv1_ln_fwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_i, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "smx_conv1_ln_bwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "smx_linear_k16_fwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_vp]),
    "smx_conv2d_s2_fwd": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_conv2d_s2_wgrad_workspace": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "smx_conv2d_s2_wgrad": (c_i, [c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "smx_conv2d_s2_dgrad": (c_i, [c_i, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "smx_axpby": (c_i, [c_i, c_f, c_vp, c_i64, c_f, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_vp]),
    "smx_dropout": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_add_rowtable": (c_i, [c_i, c_vp, c_i64, c_vp, c_i, c_i, c_i, c_vp]),
    "smx_cast_from_f32": (c_i, [c_i, c_vp, c_vp, c_i64, c_vp]),
    "smx_cast_to_f32": (c_i, [c_i, c_vp, c_vp, c_i64, c_vp]),
    "smx_adamw_step": (c_i, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_vp, c_vp, c_vp]),
    "smx_utt_meanstd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_vp]),
    "smx_stats_combine": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_vp, c_f, c_vp]),
    "smx_colnorm": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_vp]),
    "smx_log_softmax_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_vp]),
    "smx_log_softmax_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_vp]),
    "smx_ctc_workspace": (c_sz, [c_i, c_i, c_i]),
    "smx_ctc_loss_fwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp]),
    "smx_ctc_loss_bwd": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp, c_i64, c_vp,
                               c_vp]),
    "smx_get_config": (c_i, [c_vp]),
    "smx_gemm_ln_tile_rows": (c_i, []),
    "smx_gemm_ln_tile_rows_for": (c_i, [c_i, c_i]),
    "smx_gemm_panel_slabs_ok": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "smx_gemm_panel_slabs": (c_i, [c_i, c_vp, c_i64, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp]),
    "smx_slab_epilogue_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "smx_slab_epilogue": (c_i, [c_i, c_vp, c_i, c_i64, c_vp, c_i64, c_i, c_i, ctypes.POINTER(Epilogue), c_vp]),
    "smx_chunk_mean_sharded": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_i, c_i, c_vp, c_vp]),
    "smx_expdecay_mean_sharded": (c_i, [c_i, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp]),
    "smx_wgrad_group_direct_ok": (c_i, [c_i, c_i, c_i]),
    "smx_wgrad_group_direct": (c_i, [c_i, c_i, ctypes.POINTER(WgradDirectItem), c_i, c_vp]),
    "smx_layernorm_bwd2_slabs": (c_i, [c_i, c_vp, c_i, c_i64, c_vp, c_i64, c_i, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_vp,
                                       c_vp, c_i64, c_f, c_vp, c_f, ctypes.c_uint64, c_vp, c_vp]),
    "smx_gemm_panel_rows": (c_i, [c_i, c_i]),
    "smx_step_counter_add": (c_i, [c_vp, ctypes.c_uint64, c_vp]),
    "smx_stream_capture_id": (c_i, [c_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "smx_sumsq_workspace": (c_sz, []),
    "smx_sumsq": (c_i, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "smx_clip_factor": (c_i, [c_vp, c_f, c_f, c_vp, c_vp]),
}

_lib = None


def lib():
    """Load libsmx.so (once).  Fails loudly: there is no CPU / eager fallback for the product path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or bash summarymixing_amd/csrc/build.sh). summarymixing_amd has no fallback path.")
        # torch first: libsmx.so needs libamdhip64.so, and the process must end up with ONE HIP runtime - the one torch
        # ships and initialises.  Loaded the other way round (libsmx pulling /opt/rocm's copy in before `import torch`), the
        # library's launches fail with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class Config(ctypes.Structure):
    """smx_config of include/smx.h: the knobs the library read from the environment once."""
    _fields_ = [("ln_tile_rows", ctypes.c_int32),
                ("gemm_ablate", ctypes.c_int32), ("wgroup_ablate", ctypes.c_int32), ("dwroll_ablate", ctypes.c_int32),
                ("diag_build", ctypes.c_int32), ("t256", ctypes.c_int32), ("panel_rows", ctypes.c_int32), ("pool_fuse_max_rows", ctypes.c_int32), ("ln_tile64", ctypes.c_int32), ("pad_", ctypes.c_int32)]


def get_config():
    """The knobs in force as a dict (smx_get_config)."""
    c = Config()
    check(lib().smx_get_config(ctypes.byref(c)), "smx_get_config")
    return {name: getattr(c, name) for name, _ in Config._fields_}


def check(code, what):
    if code != 0:
        msg = lib().smx_last_error()
        raise RuntimeError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")
