"""Thin torch-tensor wrappers over the C-ABI of libsmx.so.

torch is plumbing here: tensors own device memory, ``data_ptr()`` / strides / the current HIP stream are
handed to the kernels.  All arithmetic happens in the hand-written gfx950 kernels.  Matrices are 2-D
views ``(rows, cols)`` with unit stride along cols; the row stride is the leading dimension.
"""
import ctypes


import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"summarymixing_amd supports float32 and bfloat16 activations, got {t.dtype}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---- in-step kernel timing (bench.py's roofline_kernels): while `_PROF` is a list every wrapper below brackets its
# launch with two HIP events on the stream it launches on (the wgrad slabs run on a side stream) and appends
# (kernel family + shape, algorithmic bytes, flops, start, end).  Off (None) in normal operation: one `is None` test.
_PROF = None


def prof_start():
    global _PROF
    _PROF = []


def prof_stop(symbols=False):
    """-> list of (name, algorithmic bytes, flops, milliseconds); synchronises the device.  symbols=True: a fifth field, the
    kernel symbol as a profiler prints it (GEMMs: the instantiation smx_gemm_plan_query reports; other launches: their family)."""
    global _PROF
    recs, _PROF = _PROF, None
    torch.cuda.synchronize()
    if symbols:
        return [(r[0], r[1], r[2], r[3].elapsed_time(r[4]), r[5] if len(r) > 5 and r[5] else r[0].split(" (")[0].split(" dW")[0]) for r in recs]
    return [(r[0], r[1], r[2], r[3].elapsed_time(r[4])) for r in recs]


def _pb(name, nbytes, flops=0.0, sym=None):
    if _PROF is None:
        return None
    st = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    return (name, float(nbytes), float(flops), e0, st, sym)


def _pe(tok):
    if tok is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(tok[4])
        _PROF.append(tok[:4] + (e1, tok[5]))


def _es(t):
    return 2 if t.dtype == torch.bfloat16 else 4


def _mat(t):
    """(ptr, ld) of a 2-D view with unit inner stride."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), f"need a (rows, cols) view with unit col stride, got {t.shape} {t.stride()}"
    assert t.is_cuda, "summarymixing_amd kernels run on the GPU only (no CPU fallback)"
    return _p(t), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def rows2d(x):
    """(B,T,D) or (N,D) -> (N,D) view without copying (rows must be uniformly strided)."""
    if x.dim() == 2:
        return x
    assert x.dim() == 3
    B, T, D = x.shape
    if x.stride(2) != 1 and D > 1:
        x = x.contiguous()
    if B > 1 and x.stride(0) != T * x.stride(1):
        x = x.contiguous()
    return x.as_strided((B * T, D), (x.stride(1), 1), x.storage_offset()) if T * B > 0 else x.reshape(B * T, D)


def epilogue(bias=None, c0=None, c0_mode=L.C0_NONE, c0_div=0, act=L.ACT_NONE, out_mode=L.OUT_T, z=None,
             row_mask=None, res=None, alpha=1.0, bias_batch_stride=0, drop=None, c0_post=False, act_grad_z=None,
             colsum=None, ln_bwd=None, ln_fwd=None, drop_cols=0, ln_fwd2=None):
    """Build an smx_epilogue.  act_grad_z: the saved pre-activation of the UPSTREAM layer (SMX_EPI_ACT_GRAD: the GEMM
    then emits alpha * D(acc * act'(z)) * mask, i.e. the upstream dZ); colsum: fp32 [M] accumulator of the output's
    column sums (the upstream bias gradient), workspace attached by gemm()."""
    e = L.Epilogue()
    e.epoch = _epoch()                # (the device step counter of the training loop, or NULL: set_step_counter below)
    e.bias = bias.data_ptr() if bias is not None else None
    e.bias_batch_stride = bias_batch_stride
    if c0 is not None:
        assert c0.dtype == torch.float32
        e.c0, e.ldc0 = _mat(c0)
        e.c0 = c0.data_ptr()
        e.c0_mode, e.c0_div = c0_mode, c0_div
    e.act, e.out_mode = act, out_mode
    if z is not None:
        e.z, e.ldz = z.data_ptr(), _mat(z)[1]
    if row_mask is not None:
        assert row_mask.dtype == torch.uint8
        e.row_mask = row_mask.data_ptr()
    if res is not None:
        e.res, e.ldr = res.data_ptr(), _mat(res)[1]
        if res.dtype == torch.float32 and out_mode == L.OUT_F32:
            e.io_flags |= L.IO_RES_F32                   # fp32 residual stream next to bf16 operands
    e.alpha = alpha
    e.flags = L.EPI_C0_POST if c0_post else 0
    if act_grad_z is not None:
        assert z is None and res is None
        e.z, e.ldz = act_grad_z.data_ptr(), _mat(act_grad_z)[1]
        e.flags |= L.EPI_ACT_GRAD
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.is_contiguous()
        e.colsum = colsum.data_ptr()
    if drop is not None and drop[0] > 0.0:
        e.drop_p, e.drop_seed = drop
        e.drop_cols = drop_cols          # > 0: dropout on the first drop_cols output columns only (mask index n*drop_cols + m)
    if ln_bwd is not None:
        # (x, stats, gamma, partial, dx2 | None, second | None): SMX_EPI_LN_BWD - the GEMM output is the gradient of LN(x)
        x, stats, gamma, partial, dx2, second = ln_bwd[:6]
        if len(ln_bwd) > 6 and ln_bwd[6] is not None:      # (beta, act): the LayerNorm had a fused activation
            e.lnf_beta, e.lnf_act = ln_bwd[6][0].data_ptr(), ln_bwd[6][1]
        e.ln_x, e.ln_ldx = x.data_ptr(), _mat(x)[1]
        if x.dtype == torch.float32 and len(ln_bwd) > 7 and ln_bwd[7]:     # (x float32 next to bf16 gradients)
            e.io_flags |= L.IO_LNX_F32
        e.ln_stats, e.ln_gamma, e.ln_partial = stats.data_ptr(), gamma.data_ptr(), partial.data_ptr()
        e.flags |= L.EPI_LN_BWD
        if dx2 is not None:
            a2, m2, drop2 = second[:3]
            if len(second) > 3 and second[3] is not None:  # (z, act): dx2 = alpha * D(dx * act'(z)) * mask
                assert z is None and act_grad_z is None
                e.z, e.ldz, e.act = second[3].data_ptr(), _mat(second[3])[1], second[4]
            e.ln_dx2, e.ln_lddx2, e.ln_alpha2 = dx2.data_ptr(), _mat(dx2)[1], a2
            e.ln_mask2 = m2.data_ptr() if m2 is not None else None
            if drop2 is not None and drop2[0] > 0.0:
                e.ln_drop_p2, e.ln_drop_seed2 = drop2
    if ln_fwd is not None:
        # (gamma, beta, y, stats | None, eps, act): SMX_EPI_LN_FWD - y = act(LN(output)) appended to the epilogue
        gamma, beta, y, stats, eps, lact = ln_fwd
        e.lnf_gamma, e.lnf_beta = gamma.data_ptr(), beta.data_ptr()
        e.lnf_y, e.lnf_ldy = y.data_ptr(), _mat(y)[1]
        if y.dtype == torch.float32 and out_mode == L.OUT_F32 and res is not None and res.dtype == torch.float32:
            e.io_flags |= L.IO_LNFY_F32                  # the LayerNorm output is the fp32 residual stream itself
        e.lnf_stats = stats.data_ptr() if stats is not None else None
        e.lnf_eps, e.lnf_act = eps, lact
        e.flags |= L.EPI_LN_FWD
    if ln_fwd2 is not None:
        # (gamma2, beta2, y2, stats2 | None, eps2): a SECOND LayerNorm of the first one's output (smx_epilogue.lnf2_*; check
        # smx_gemm_ln_pair_ok first)
        assert ln_fwd is not None
        gamma2, beta2, y2, stats2, eps2 = ln_fwd2
        e.lnf2_gamma, e.lnf2_beta = gamma2.data_ptr(), beta2.data_ptr()
        e.lnf2_y, e.lnf2_ldy = y2.data_ptr(), _mat(y2)[1]
        e.lnf2_stats = stats2.data_ptr() if stats2 is not None else None
        e.lnf2_eps = eps2
    return e


def gemm_symbol(layout, a, b, c, N, M, K, epi=None, batch=1, sa=0, sb=0, sc=0, splits=1, lda=None, ldb=None, ldc=None):
    """The kernel symbol smx_gemm launches for these arguments, spelled as rocprofv3 prints it (smx_gemm_plan_query: the same
    checks and dispatch as the launch itself, nothing launched)."""
    pa, la = _mat(a)
    pb, lb = _mat(b)
    pc, lc = _mat(c)
    plan = L.GemmPlan()
    L.check(L.lib().smx_gemm_plan_query(layout, dt(a), pa, lda or la, sa, pb, ldb or lb, sb, pc, ldc or lc, sc, N, M, K, batch, splits,
                                        ctypes.byref(epi if epi is not None else epilogue()), ctypes.byref(plan)), "smx_gemm_plan_query")
    if plan.kernel == 1:
        return "gemm_tn_dma_kernel"
    if plan.kernel != 0:
        return None
    tb = lambda v: "true" if v else "false"
    return (f"gemm_kernel<{'bf16_t' if a.dtype == torch.bfloat16 else 'float'}, {tb(plan.a_kc)}, {tb(plan.b_kc)}, {plan.tile_n}, "
            f"{plan.tile_m}, {tb(plan.vec)}, {plan.lnf}, {plan.gather}>")


def gemm(layout, a, b, c, N, M, K, epi=None, batch=1, sa=0, sb=0, sc=0, splits=1, lda=None, ldb=None, ldc=None):
    """C (N x M) = epi(op(A) op(B)); a/b/c are 2-D views of batch 0 (batch strides in elements)."""
    pa, la = _mat(a)
    pb, lb = _mat(b)
    pc, lc = _mat(c)
    if epi is None:
        epi = epilogue()
    assert a.dtype == b.dtype
    if epi.colsum:
        epi.workspace = _workspace(L.lib().smx_gemm_colsum_workspace(N, M), c.device, "gemm_colsum").data_ptr()
    tok = None
    if _PROF is not None:
        es, osz = _es(a), (4 if epi.out_mode == L.OUT_F32 else _es(c))
        # algorithmic bytes of the side tensors: residual (float32 on the fp32 residual stream), saved / re-read pre-activation,
        # the LayerNorm input of a fused backward (float32 on the fp32 stream), its second output, the normalised copy of a
        # fused forward (float32 when it is the stream itself)
        side = (4 if epi.io_flags & L.IO_RES_F32 else es) * (1 if epi.res else 0) + es * (1 if epi.z else 0)
        side += (4 if epi.io_flags & L.IO_LNX_F32 else es) * (1 if epi.flags & L.EPI_LN_BWD else 0) + es * (1 if epi.ln_dx2 else 0)
        side += (4 if epi.io_flags & L.IO_LNFY_F32 else es) * (1 if epi.flags & L.EPI_LN_FWD else 0)
        nb = batch * ((N * K + M * K) * es + side * N * M + N * M * osz)
        tag = ("+LNbwd" if epi.flags & L.EPI_LN_BWD else "") + ("+LNfwd" if epi.flags & L.EPI_LN_FWD else "")
        tag += "".join(t for t, on in (("+bias", epi.bias), ("+act", epi.act != L.ACT_NONE and not (epi.flags & L.EPI_ACT_GRAD)),
                                      ("+Z", epi.z and not (epi.flags & L.EPI_ACT_GRAD)), ("+actgrad(z)", epi.flags & L.EPI_ACT_GRAD),
                                      ("+res", epi.res), ("+c0", epi.c0), ("+mask", epi.row_mask), ("+drop", epi.drop_p > 0)) if on)
        tok = _pb(f"gemm {('NT', 'NN', 'TN')[layout]} {'bf16' if es == 2 else 'f32'} ({N}x{K})x({K}x{M}){'' if batch == 1 else ' x%d' % batch} {tag}",
                  nb, 2.0 * N * M * K * batch,
                  gemm_symbol(layout, a, b, c, N, M, K, epi, batch, sa, sb, sc, splits, lda, ldb, ldc))
    L.check(L.lib().smx_gemm(layout, dt(a), pa, lda or la, sa, pb, ldb or lb, sb, pc, ldc or lc, sc, N, M, K, batch,
                             splits, ctypes.byref(epi), _stream()), "smx_gemm")
    _pe(tok)
    return c


def weight_pack(W, transposed=False, bias=None, out=None):
    """MFMA-fragment-order image of a bf16 weight (+ its fp32 bias) for gemm_panel (smx_weight_pack).  W: (M, K) [transposed=False: the
    weight of a forward Linear] or (K, M) [transposed=True: the weight of the Linear whose dgrad dH = dY W is computed]."""
    assert W.dtype == torch.bfloat16 and W.dim() == 2
    M, K = (W.shape[1], W.shape[0]) if transposed else (W.shape[0], W.shape[1])
    if out is None:
        out = torch.empty((L.lib().smx_weight_pack_bytes(M, K) // 2,), dtype=torch.bfloat16, device=W.device)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == M
    pw, lw = _mat(W)
    tok = _pb(f"weight_pack ({M}x{K}){' T' if transposed else ''}", 4.0 * M * K)
    L.check(L.lib().smx_weight_pack(L.BF16, pw, lw, 1 if transposed else 0, _p(bias), M, K, _p(out), _stream()), "smx_weight_pack")
    _pe(tok)
    return out


def weight_pack_jobs(jobs_dev, njobs, total_blocks, nelem=0):
    """Every job of a device table of smx_pack_job in one launch (smx_weight_pack_jobs)."""
    tok = _pb(f"weight_pack_jobs ({njobs} weights)", 4.0 * nelem)
    L.check(L.lib().smx_weight_pack_jobs(L.BF16, _p(jobs_dev), njobs, total_blocks, _stream()), "smx_weight_pack_jobs")
    _pe(tok)


def gemm_panel_ok(a, M, K):
    return a.dtype == torch.bfloat16 and L.lib().smx_gemm_panel_ok(L.BF16, a.shape[0], M, K) == 1


def gemm_panel(a, wp, c, N, M, K, epi=None):
    """C (N x M) = epi(A Wp) on the panel-resident kernel (smx_gemm_panel): A (N, K) bf16, wp = weight_pack(...)."""
    pa, la = _mat(a)
    pc, lc = _mat(c)
    if epi is None:
        epi = epilogue()
    tok = None
    if _PROF is not None:
        ag = bool(epi.flags & L.EPI_ACT_GRAD)
        nb = (N * K + M * K) * 2 + (2 if epi.z else 0) * N * M + N * M * 2
        tag = "".join(t for t, on in (("+act", epi.act != L.ACT_NONE and not ag), ("+Z", epi.z and not ag),
                                      ("+actgrad(z)", ag), ("+drop", epi.drop_p > 0)) if on)
        tok = _pb(f"gemm panel bf16 ({N}x{K})x({K}x{M}) {tag}", nb, 2.0 * N * M * K,
                  f"gemm_panel_kernel<{K}, {1 if ag else 0}, {epi.act}, {L.lib().smx_gemm_panel_rows(N, M)}>")
    L.check(L.lib().smx_gemm_panel(L.BF16, pa, la, _p(wp), pc, lc, N, M, K, ctypes.byref(epi), _stream()), "smx_gemm_panel")
    _pe(tok)
    return c


def panel_slabs_ok(a, M, K, nslice):
    return a.dtype == torch.bfloat16 and L.lib().smx_gemm_panel_slabs_ok(L.BF16, a.shape[0], M, K, nslice) == 1


def gemm_panel_slabs(a, wp, slabs, N, M, K, nslice):
    """slabs[s] (N x M, fp32) = A[:, s K:(s + 1) K] W_s^T (smx_gemm_panel_slabs): A (N, nslice K) bf16, wp = nslice consecutive
    weight_pack images (weight_pack_slices)."""
    pa, la = _mat(a)
    tok = _pb(f"gemm panel slabs bf16 ({N}x{nslice * K})x({nslice * K}x{M}) S={nslice}", (N * nslice * K + M * nslice * K) * 2 + nslice * N * M * 4,
              2.0 * N * M * K * nslice, f"gemm_panel_kernel<{K}, 2, 0, *>")
    L.check(L.lib().smx_gemm_panel_slabs(L.BF16, pa, la, _p(wp), _p(slabs), N, M, K, nslice, _stream()), "smx_gemm_panel_slabs")
    _pe(tok)
    return slabs


def weight_pack_slices(W, K, transposed=False, out=None):
    """The K-slices of a bf16 weight packed one after the other (no bias) for gemm_panel_slabs.  W (M, nslice K) [transposed=False:
    a forward Linear's weight, slices = column ranges] or (nslice K, M) [transposed=True: the same weight seen from its dgrad,
    slices = row ranges]."""
    Ktot, M = (W.shape[0], W.shape[1]) if transposed else (W.shape[1], W.shape[0])
    ns = Ktot // K
    assert ns * K == Ktot
    per = L.lib().smx_weight_pack_bytes(M, K) // 2
    if out is None:
        out = torch.empty((ns * per,), dtype=torch.bfloat16, device=W.device)
    for s_ in range(ns):
        Ws = W[s_ * K:(s_ + 1) * K] if transposed else W[:, s_ * K:(s_ + 1) * K]
        L.check(L.lib().smx_weight_pack(L.BF16, _p(Ws), Ws.stride(0), 1 if transposed else 0, None, M, K,
                                        ctypes.c_void_p(out.data_ptr() + 2 * s_ * per), _stream()), "smx_weight_pack")
    return out


def slab_epilogue(slabs, nslab, c, N, M, epi):
    """c = epilogue(sum of the float32 slabs) + the LayerNorm(s) the epilogue names (smx_slab_epilogue)."""
    pc, lc = _mat(c)
    tok = _pb(f"slab epilogue ({N}x{M}) S={nslab}{'+LNfwd' if epi.flags & L.EPI_LN_FWD else ''}",
              N * M * (4 * nslab + c.element_size() + (4 if epi.io_flags & L.IO_RES_F32 else 2) * (1 if epi.res else 0) + 2 * (1 if epi.z else 0) +
                       ((4 if epi.io_flags & L.IO_LNFY_F32 else 2) if epi.flags & L.EPI_LN_FWD else 0) + (2 if epi.lnf2_y else 0)))
    L.check(L.lib().smx_slab_epilogue(L.BF16, _p(slabs), nslab, N * M, pc, lc, N, M, ctypes.byref(epi), _stream()), "smx_slab_epilogue")
    _pe(tok)
    return c


def wgrad(dz, x, gW, rows, M, K, batch=1, sz=0, sx=0, sw=0, lddz=None, ldx=None, lddw=None, alpha=1.0, dbias=None):
    """gW[b] (M x K) += alpha * dz[b]^T x[b]  (slab split-K + fixed-order reduction; see smx_linear_wgrad).
    dbias (fp32 (batch, M), contiguous): += alpha * column sums of dz from the same launch (needs K % 4 == 0)."""
    pz, lz = _mat(dz)
    px, lx = _mat(x)
    pw, lw = _mat(gW)
    assert gW.dtype == torch.float32
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == batch * M and K % 4 == 0
    ws = _workspace(L.lib().smx_linear_wgrad_workspace(rows, M, K, batch), dz.device, slot=1)
    tok = _pb(f"wgrad(+reduce) {'bf16' if _es(dz) == 2 else 'f32'} dW({M}x{K}) over {rows} frames{'' if batch == 1 else ' x%d' % batch}",
              batch * ((M + K) * rows * _es(dz) + 4 * M * K), 2.0 * rows * M * K * batch) if _PROF is not None else None
    L.check(L.lib().smx_linear_wgrad(dt(dz), pz, lddz or lz, sz, px, ldx or lx, sx, pw, lddw or lw, sw, _p(dbias), rows, M, K,
                                     batch, alpha, _p(ws), _stream()), "smx_linear_wgrad")
    _pe(tok)


def wgrad_partial(dz, x, rows, M, K, ws, batch=1, sz=0, sx=0, lddz=None, ldx=None, want_bias=False):
    """Only the split-K slabs of a wgrad (and the bias partials behind them) into the caller-owned workspace `ws`.
    -> (nslabs, slab_stride, bias_offset) in floats; reduce later with reduce_jobs."""
    pz, lz = _mat(dz)
    px, lx = _mat(x)
    n, st, bo = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
    tok = _pb(f"wgrad slabs {'bf16' if _es(dz) == 2 else 'f32'} dW({M}x{K}) over {rows} frames{'' if batch == 1 else ' x%d' % batch}",
              batch * ((M + K) * rows * _es(dz) + 4 * M * K), 2.0 * rows * M * K * batch) if _PROF is not None else None
    L.check(L.lib().smx_linear_wgrad_partial(dt(dz), pz, lddz or lz, sz, px, ldx or lx, sx, rows, M, K, batch,
                                             1 if want_bias else 0, _p(ws), ctypes.byref(n), ctypes.byref(st),
                                             ctypes.byref(bo), _stream()), "smx_linear_wgrad_partial")
    _pe(tok)
    return n.value, st.value, bo.value


def wgrad_group(items, nitems, rows, splits):
    """One launch for all the weight gradients of a layer (smx_wgrad_group): `items` is a ctypes array of L.WgradItem with
    workspaces attached; the slabs / bias partials stay in those workspaces for reduce_jobs."""
    tok = None
    if _PROF is not None:
        nb = sum((items[i].M + items[i].K) * rows * 2 + 4 * items[i].M * items[i].K for i in range(nitems))
        fl = sum(2.0 * rows * items[i].M * items[i].K for i in range(nitems))
        tok = _pb(f"wgrad_group bf16 ({nitems} weights: " + " ".join(f"{items[i].M}x{items[i].K}" for i in range(nitems)) +
                  f") over {rows} frames", nb, fl, "wgrad_group_kernel<32>")
    L.check(L.lib().smx_wgrad_group(L.BF16, rows, items, nitems, splits, _stream()), "smx_wgrad_group")
    _pe(tok)


def wgrad_group_direct(recs, rows):
    """All the weight gradients of a layer in one launch, added INTO the gradients (smx_wgrad_group_direct, small batches):
    recs = [(dz, x, gW, dbias | None, M, K), ...] with gW (M, K) float32 views."""
    n = len(recs)
    items = (L.WgradDirectItem * n)()
    for it, (dz, x, gW, dbias, M, K) in zip(items, recs):
        it.dZ, it.lddz, it.X, it.ldx = dz.data_ptr(), dz.stride(0), x.data_ptr(), x.stride(0)
        it.dW, it.lddw, it.dbias, it.M, it.K = gW.data_ptr(), gW.stride(0), (dbias.data_ptr() if dbias is not None else None), M, K
    tok = None
    if _PROF is not None:
        nb = sum((M + K) * rows * 2 + 8 * M * K for _, _, _, _, M, K in recs)
        fl = sum(2.0 * rows * M * K for _, _, _, _, M, K in recs)
        tok = _pb(f"wgrad_group direct bf16 ({n} weights: " + " ".join(f"{M}x{K}" for _, _, _, _, M, K in recs) + f") over {rows} frames", nb, fl,
                  "wgrad_group_direct_kernel")
    L.check(L.lib().smx_wgrad_group_direct(L.BF16, rows, items, n, _stream()), "smx_wgrad_group_direct")
    _pe(tok)


def reduce_jobs(jobs_dev, starts_dev, njobs, total_blocks, nbytes=0):
    tok = _pb(f"reduce_jobs ({njobs} jobs)", nbytes)
    L.check(L.lib().smx_reduce_jobs(_p(jobs_dev), _p(starts_dev), njobs, total_blocks, _stream()), "smx_reduce_jobs")
    _pe(tok)


def act_mask_bwd(dy, z, mask, act, alpha=1.0, dz=None, dbias=None, dgroup=None, gdiv=0, drop=None):
    N, M = dy.shape
    pdy, lddy = _mat(dy)
    pz, ldz = (_mat(z) if z is not None else (None, 0))
    pdz, lddz = (_mat(dz) if dz is not None else (None, 0))
    pg, ldg = (_mat(dgroup) if dgroup is not None else (None, 0))
    ws = _workspace(L.lib().smx_act_mask_bwd_workspace(N, M), dy.device, slot=3) if dbias is not None else None
    dp, ds = (drop if drop is not None else (0.0, 0))
    tok = _pb(f"act_mask_bwd ({N}x{M})", (1 + (z is not None) + (dz is not None)) * N * M * _es(dy))
    L.check(L.lib().smx_act_mask_bwd(dt(dy), pdy, lddy, pz, ldz, _p(mask), pdz, lddz, N, M, act, alpha, _p(dbias), pg,
                                     ldg, gdiv, dp, ds, _epoch(), _p(ws), _stream()), "smx_act_mask_bwd")
    _pe(tok)
    return dz


_ws_cache = {}


def _workspace(nbytes, device, slot=0):
    key = (device, torch.cuda.current_stream().cuda_stream, slot)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def masked_mean(s, mask, B, T, scale=True, want_inv=False):
    """s: (B*T, D) view; mask: (B*T) uint8 or None -> (out (B,D) fp32, inv_count (B) fp32 | None)."""
    D = s.shape[1]
    ps, lds = _mat(s)
    out = torch.empty((B, D), dtype=torch.float32, device=s.device)
    inv = torch.empty((B,), dtype=torch.float32, device=s.device) if want_inv else None
    ws = _workspace(L.lib().smx_masked_mean_workspace(B, T, D), s.device)
    tok = _pb(f"masked_mean pool ({B},{T},{D})", B * T * D * _es(s) + B * T + 4 * B * D)
    L.check(L.lib().smx_masked_mean_fwd(dt(s), ps, lds, _p(mask), _p(out), _p(inv), B, T, D, 1 if scale else 0, _p(ws),
                                        _stream()), "smx_masked_mean_fwd")
    _pe(tok)
    return out, inv


def bcast_rows(g, inv, ds, B, T, drop=None):
    """ds[b*T+t, :] = D(g[b,:] * inv[b])   (D: optional fused dropout (p, seed), index row * D + col)"""
    pds, ldds = _mat(ds)
    dp, dseed = drop if drop is not None else (0.0, 0)
    tok = _pb(f"bcast_rows ({B},{T},{ds.shape[1]})", B * T * ds.shape[1] * _es(ds))
    L.check(L.lib().smx_masked_mean_bwd(dt(ds), _p(g), _p(inv), pds, ldds, B, T, ds.shape[1], dp, dseed, _epoch(), _stream()),
            "smx_masked_mean_bwd")
    _pe(tok)
    return ds


def pool_bcast_ok(B, T, D):
    return L.lib().smx_pool_bcast_ok(B, T, D) == 1


def pool_bcast(s, mask_in, B, T, ds=None, scale=True, want_mean=True, want_inv=False, inv_in=None, drop=None, z=None, mask_out=None,
               act=L.ACT_NONE):
    """The masked mean over time and its broadcast in ONE launch (smx_pool_bcast; small batches: pool_bcast_ok).
    -> (mean (B, D) fp32 | None, inv_count (B) fp32 | None); ds (B*T, D) receives D(value * inv_in) [* act'(z) * mask_out]."""
    D = s.shape[1]
    ps, lds_ = _mat(s)
    mean = torch.empty((B, D), dtype=torch.float32, device=s.device) if want_mean else None
    inv = torch.empty((B,), dtype=torch.float32, device=s.device) if want_inv else None
    pds, ldds = _mat(ds) if ds is not None else (None, 0)
    pz, ldz = _mat(z) if z is not None else (None, 0)
    dp, dseed = drop if drop is not None else (0.0, 0)
    nb = B * T * D * _es(s) * (1 + (ds is not None) + (z is not None)) + B * T
    tok = _pb(f"pool+bcast ({B},{T},{D})", nb)
    L.check(L.lib().smx_pool_bcast(dt(s), ps, lds_, _p(mask_in), _p(mean), _p(inv_in), _p(inv), pds, ldds, B, T, D, 1 if scale else 0,
                                   dp, dseed, _epoch() if dp > 0 else None, pz, ldz, _p(mask_out), act, _stream()), "smx_pool_bcast")
    _pe(tok)
    return mean, inv


def bcast_rows_act_bwd(g, inv, ds, B, T, z, mask, act):
    """ds[b*T+t, :] = g[b,:] * inv[b] * act'(z[b*T+t, :]) * mask[b*T+t]   (z and / or mask may be None)"""
    pds, ldds = _mat(ds)
    pz, ldz = (_mat(z) if z is not None else (None, 0))
    tok = _pb(f"bcast_rows+act_bwd ({B},{T},{ds.shape[1]})", (1 + (z is not None)) * B * T * ds.shape[1] * _es(ds))
    L.check(L.lib().smx_masked_mean_bwd_act(dt(ds), _p(g), _p(inv), pds, ldds, pz, ldz, _p(mask), act, B, T, ds.shape[1],
                                            _stream()), "smx_masked_mean_bwd_act")
    _pe(tok)
    return ds


def chunk_mean(s, out, B, T, chunk, left, reverse=False):
    D = s.shape[1]
    ps, lds = _mat(s)
    po, ldo = _mat(out)
    ws = _workspace(L.lib().smx_chunk_mean_workspace(B, T, D, chunk), s.device)
    fn = L.lib().smx_chunk_mean_bwd if reverse else L.lib().smx_chunk_mean_fwd
    L.check(fn(dt(s), ps, lds, po, ldo, B, T, D, chunk, -1 if left is None else left, _p(ws), _stream()),
            "smx_chunk_mean")
    return out


def expdecay_mean(s, out, B, T, decay, reverse=False):
    """out = (M s)/rowsum(M) with M_ij = decay^|i-j| (reverse: the transposed operator M (s/rowsum(M))); O(T)."""
    D = s.shape[1]
    ps, lds = _mat(s)
    po, ldo = _mat(out)
    ws = _workspace(L.lib().smx_expdecay_mean_workspace(B, T, D), s.device, slot=5)
    fn = L.lib().smx_expdecay_mean_bwd if reverse else L.lib().smx_expdecay_mean_fwd
    L.check(fn(dt(s), ps, lds, po, ldo, B, T, D, float(decay), _p(ws), _stream()), "smx_expdecay_mean")
    return out


def chunk_mean_sharded(x, out, B, T, D, chunk, left, reverse, c_off, phase, ws, carry=None, carry_c0=0, carry_n=0):
    """One phase of the DynChunk window mean on a sequence-parallel shard (smx_chunk_mean_sharded): phase 1 = chunk sums of x into ws
    ((B, T // chunk, D) float32), phase 2 = window combine (+ carry) into out."""
    px, ldx = _mat(x) if x is not None else (None, 0)
    po, ldo = _mat(out) if out is not None else (None, 0)
    t = x if x is not None else out
    L.check(L.lib().smx_chunk_mean_sharded(dt(t), px, ldx, po, ldo, B, T, D, chunk, -1 if left is None else left, 1 if reverse else 0, c_off,
                                           phase, _p(carry), carry_c0, carry_n, _p(ws), _stream()), "smx_chunk_mean_sharded")


def expdecay_mean_sharded(x, out, B, T, decay, reverse, t_off, T_glob, phase, ends, ws):
    """One phase of the expdecay summary on a sequence-parallel shard (smx_expdecay_mean_sharded): phase 1 fills ends (2, B, D) with the
    states leaving the shard, phase 2 takes the states entering it from `ends` and writes out."""
    D = x.shape[1]
    px, ldx = _mat(x)
    po, ldo = _mat(out) if out is not None else (None, 0)
    L.check(L.lib().smx_expdecay_mean_sharded(dt(x), px, ldx, po, ldo, B, T, D, float(decay), 1 if reverse else 0, t_off, T_glob, phase,
                                              _p(ends), _p(ws), _stream()), "smx_expdecay_mean_sharded")


def layernorm_fwd(x, gamma, beta, eps, want_stats, act=L.ACT_NONE, out_dtype=None):
    """out_dtype (default x.dtype): bf16 output of a float32 input = LayerNorm of the fp32 residual stream feeding a bf16 GEMM
    (smx_layernorm_fwd_x32)."""
    N, D = x.shape
    out_dtype = out_dtype or x.dtype
    y = torch.empty((N, D), dtype=out_dtype, device=x.device)
    stats = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_stats else None
    px, ldx = _mat(x)
    tok = _pb(f"layernorm_fwd ({N}x{D})", N * D * (_es(x) + _es(y)))
    if out_dtype != x.dtype:
        assert x.dtype == torch.float32, "mixed LayerNorm: float32 in, compute dtype out"
        L.check(L.lib().smx_layernorm_fwd_x32(dt(y), px, ldx, _p(gamma), _p(beta), _p(y), D, _p(stats), N, D, eps, act, _stream()),
                "smx_layernorm_fwd_x32")
    else:
        L.check(L.lib().smx_layernorm_fwd(dt(x), px, ldx, _p(gamma), _p(beta), _p(y), D, _p(stats), N, D, eps, act, _stream()),
                "smx_layernorm_fwd")
    _pe(tok)
    return y, stats


def layernorm_pair_ok(x, out_dtype, *affine):
    """Can smx_layernorm_fwd_pair_x32 take this float32 stream tensor - and these gamma / beta vectors?  The kernel reads all four
    with 16-byte loads and returns SMX_EUNSUPPORTED for an unaligned one (a parameter that is a view at an odd offset of a flat
    buffer): checked here so that the caller takes the two-launch path instead of failing."""
    return (x.dtype == torch.float32 and x.is_cuda and x.shape[1] % 4 == 0 and x.shape[1] <= 2048 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and out_dtype in (torch.bfloat16, torch.float32)
            and all(v.is_contiguous() and v.data_ptr() % 16 == 0 for v in affine))


def layernorm_fwd_pair(x, gamma1, beta1, eps1, gamma2, beta2, eps2, want_stats, out_dtype):
    """(y1, stats1, y2, stats2): y1 = LN1(x) float32, y2 = LN2(y1) in out_dtype, one pass over the float32 stream x
    (smx_layernorm_fwd_pair_x32: the layer-final norm2 + the next layer's first LayerNorm)."""
    N, D = x.shape
    y1 = torch.empty((N, D), dtype=torch.float32, device=x.device)
    y2 = torch.empty((N, D), dtype=out_dtype, device=x.device)
    st1 = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_stats else None
    st2 = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_stats else None
    px, ldx = _mat(x)
    tok = _pb(f"layernorm_fwd_pair ({N}x{D})", N * D * (4 + 4 + _es(y2)))
    L.check(L.lib().smx_layernorm_fwd_pair_x32(dt(y2), px, ldx, _p(gamma1), _p(beta1), eps1, _p(y1), D, _p(st1), _p(gamma2), _p(beta2),
                                               eps2, _p(y2), D, _p(st2), N, D, _stream()), "smx_layernorm_fwd_pair_x32")
    _pe(tok)
    return y1, st1, y2, st2


class Slabs:
    """The float32 split-K partial products of a Linear (gemm_panel_slabs): `t` (S, N, M), to be summed by the consumer -
    slab_epilogue (forward) or layernorm_bwd (the dgrad of the Linear behind a LayerNorm)."""
    def __init__(self, t):
        self.t, self.S, self.N, self.M = t, t.shape[0], t.shape[1], t.shape[2]
        self.shape, self.dtype, self.device = (self.N, self.M), torch.bfloat16, t.device

    def sum(self, dtype=torch.bfloat16):
        """The reduced tensor by plain torch (tests / fallbacks only: the product path fuses the sum into its consumer)."""
        return self.t.sum(0).to(dtype)


def layernorm_bwd(dy, x, gamma, beta, stats, dgamma, dbeta, res=None, act=L.ACT_NONE, ws=None, dx_out=None, second=None):
    """ws: caller-owned workspace; with dgamma = dbeta = None the partial rows stay in it for a deferred reduce_jobs.
    dx_out: optional (N, D) destination view (e.g. a column slice of a wider buffer).
    second = (alpha, row_mask_u8 | None, (p, seed) | None): also return dx2 = alpha * D(dx) * mask (smx_layernorm_bwd2)."""
    N, D = x.shape
    gdt = dy.dtype                                       # gradient dtype; x may be float32 next to bf16 gradients (fp32 residual stream)
    dx = dx_out if dx_out is not None else torch.empty((N, D), dtype=gdt, device=x.device)
    if isinstance(dy, Slabs):
        # the incoming gradient is the sum of float32 split-K slabs: reducer + LayerNorm backward in one launch (smx_layernorm_bwd2_slabs)
        assert dgamma is None and dbeta is None and ws is not None and dy.N == N and dy.M == D
        px, ldx = _mat(x)
        pr, ldr = (_mat(res) if res is not None else (None, 0))
        dx2, a2, m2, dp2, ds2 = None, 1.0, None, 0.0, 0
        if second is not None:
            a2, m2, drop2 = second
            dp2, ds2 = drop2 if (drop2 is not None and drop2[0] > 0.0) else (0.0, 0)
            dx2 = torch.empty((N, D), dtype=gdt, device=x.device)
        tok = _pb(f"layernorm_bwd from slabs ({N}x{D}) S={dy.S}{'+res' if res is not None else ''}{'+2nd' if second is not None else ''}",
                  ((1 + (res is not None) + (second is not None)) * 2 + _es(x) + 4 * dy.S) * N * D)
        L.check(L.lib().smx_layernorm_bwd2_slabs(L.BF16, _p(dy.t), dy.S, N * D, px, ldx, 1 if x.dtype == torch.float32 else 0, _p(gamma), _p(beta), act,
                                                 _p(stats), pr, ldr, _p(dx), _mat(dx)[1], N, D, _p(ws), _p(dx2), D if dx2 is not None else 0, a2, _p(m2),
                                                 dp2, ds2, _epoch(), _stream()), "smx_layernorm_bwd2_slabs")
        _pe(tok)
        return dx if second is None else (dx, dx2)
    pdy, lddy = _mat(dy)
    px, ldx = _mat(x)
    pr, ldr = (_mat(res) if res is not None else (None, 0))
    if ws is None:
        ws = _workspace(L.lib().smx_layernorm_bwd_workspace(N, D), x.device, slot=2)
    dx2 = None
    a2, m2, dp2, ds2 = 1.0, None, 0.0, 0
    if second is not None:
        a2, m2, drop2 = second
        dp2, ds2 = drop2 if (drop2 is not None and drop2[0] > 0.0) else (0.0, 0)
        dx2 = torch.empty((N, D), dtype=gdt, device=x.device)
    tok = _pb(f"layernorm_bwd ({N}x{D}){'+res' if res is not None else ''}{'+2nd' if second is not None else ''}",
              ((2 + (res is not None) + (second is not None)) * _es(dy) + _es(x)) * N * D)
    fn = L.lib().smx_layernorm_bwd2 if x.dtype == gdt else L.lib().smx_layernorm_bwd2_x32
    assert x.dtype == gdt or x.dtype == torch.float32
    L.check(fn(dt(dy), pdy, lddy, px, ldx, _p(gamma), _p(beta), act, _p(stats), pr, ldr, _p(dx), _mat(dx)[1],
               _p(dgamma), _p(dbeta), N, D, _p(ws), _p(dx2), D if dx2 is not None else 0, a2, _p(m2), dp2,
               ds2, _epoch(), _stream()), "smx_layernorm_bwd")
    _pe(tok)
    return dx if second is None else (dx, dx2)


def layernorm_bwd_preact_ok(dy, x, z, dx):
    """Can smx_layernorm_bwd_preact take these views?  (bf16, D <= 2048, D % 8 == 0, 16-byte aligned rows)"""
    D = x.shape[1]
    ok = lambda t: t.dtype == torch.bfloat16 and t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) == 1
    return D <= 2048 and D % 8 == 0 and all(ok(t) for t in (dy, x, z, dx))


def layernorm_bwd_preact(dy, x, gamma, beta, stats, z, zact, dgamma, dbeta, act=L.ACT_NONE, ws=None, dx_out=None):
    """dZ = zact'(z) * LayerNorm-backward(dy) for a LayerNorm whose input is x = zact(z) (smx_layernorm_bwd_preact)."""
    N, D = x.shape
    dx = dx_out if dx_out is not None else torch.empty((N, D), dtype=dy.dtype, device=x.device)
    if ws is None:
        ws = _workspace(L.lib().smx_layernorm_bwd_workspace(N, D), x.device, slot=2)
    tok = _pb(f"layernorm_bwd+preact ({N}x{D})", 4 * _es(dy) * N * D)
    L.check(L.lib().smx_layernorm_bwd_preact(dt(dy), _p(dy), _mat(dy)[1], _p(x), _mat(x)[1], _p(gamma), _p(beta), act, _p(stats),
                                             _p(z), _mat(z)[1], zact, _p(dx), _mat(dx)[1], _p(dgamma), _p(dbeta), N, D, _p(ws),
                                             _stream()), "smx_layernorm_bwd_preact")
    _pe(tok)
    return dx


def dwconv_fwd(p, w, bias, B, T, D, k, glu, pad_mode=L.PAD_ZERO, chunk=0, gate=None, drop=None):
    """drop = (p, seed): inverted dropout of the output (the CSGU's own), fused where the kernel can (rolling CSGU path),
    else a separate in-place smx_dropout with the same mask."""
    y = torch.empty((B * T, D), dtype=p.dtype, device=p.device)
    pp, ldp = _mat(p)
    pg, ldg = (_mat(gate) if gate is not None else (None, 0))
    tok = _pb(f"dwconv_fwd ({B},{T},{D}) k={k}", 3 * B * T * D * _es(p))
    fused = False
    if drop is not None and drop[0] > 0.0 and _CSGU_DROP_FUSE:
        fused = L.lib().smx_dwconv1d_glu_fwd_drop(dt(p), pp, ldp, _p(w), _p(bias), pg, ldg, _p(y), D, B, T, D, k, 1 if glu else 0,
                                                  pad_mode, chunk, drop[0], drop[1], _epoch(), _stream()) == 0
    if not fused:
        L.check(L.lib().smx_dwconv1d_glu_fwd(dt(p), pp, ldp, _p(w), _p(bias), pg, ldg, _p(y), D, B, T, D, k, 1 if glu else 0,
                                             pad_mode, chunk, _stream()), "smx_dwconv1d_glu_fwd")
    _pe(tok)
    if drop is not None and drop[0] > 0.0 and not fused:
        dropout(y, drop[0], drop[1], out=y)
    return y


def dwconv_bwd(dy, p, w, bias, dw, dbias, B, T, D, k, glu, pad_mode=L.PAD_ZERO, chunk=0, gate=None, dgate_out=None, ws=None):
    """ws (caller-owned, smx_dwconv1d_glu_bwd_workspace bytes): the tap / bias partial rows stay in it for a deferred
    reduce_jobs and dw / dbias are not touched; then returns (dp, dgate, deferred) - deferred False means the shape is outside
    the partial-row path and the gradients were accumulated into dw / dbias directly."""
    dp = torch.empty((B * T, p.shape[1]), dtype=p.dtype, device=p.device)
    dgate = None
    if gate is not None:
        dgate = dgate_out if dgate_out is not None else torch.empty((B * T, D), dtype=p.dtype, device=p.device)
    pdy, lddy = _mat(dy)
    pp, ldp = _mat(p)
    pg, ldg = (_mat(gate) if gate is not None else (None, 0))
    tok = _pb(f"dwconv_bwd ({B},{T},{D}) k={k}", 5 * B * T * D * _es(p))

    def call(pdw, pdb, wsp):
        return L.lib().smx_dwconv1d_glu_bwd(dt(p), pdy, lddy, pp, ldp, _p(w), _p(bias), pg, ldg, _p(dp), dp.shape[1], _p(dgate),
                                            (_mat(dgate)[1] if dgate is not None else 0), pdw, pdb, B, T, D, k, 1 if glu else 0,
                                            pad_mode, chunk, wsp, _stream())
    deferred = False
    if ws is not None:
        deferred = call(None, None, _p(ws)) == 0               # (SMX_EUNSUPPORTED: not the k = 31 vector path)
    if not deferred:
        own = _workspace(L.lib().smx_dwconv1d_glu_bwd_workspace(B, T, D, k), p.device, slot=4)
        L.check(call(_p(dw), _p(dbias), _p(own)), "smx_dwconv1d_glu_bwd")
    _pe(tok)
    return (dp, dgate, deferred) if ws is not None else (dp, dgate)


def axpby(a, x, b=0.0, y0=None, out=None):
    N, D = x.shape
    if out is None:
        out = torch.empty((N, D), dtype=x.dtype, device=x.device)
    px, ldx = _mat(x)
    py0, ldy0 = (_mat(y0) if y0 is not None else (None, 0))
    po, ldo = _mat(out)
    L.check(L.lib().smx_axpby(dt(x), a, px, ldx, b, py0, ldy0, po, ldo, N, D, _stream()), "smx_axpby")
    return out


_drop_state = {"counter": 0}


def new_dropout_seed():
    """A fresh 64-bit seed per dropout site and call, derived from torch's global seed and a call counter (no
    device sync)."""
    _drop_state["counter"] += 1
    # salt: 0, or rank + 1 of a sequence-parallel shard (set_seed_salt) - the fused masks are indexed by the LOCAL frame row, so every
    # shard must draw from a seed of its own; forward and backward of a site use the same returned value either way
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _drop_state["counter"] * 0xD1B54A32D192ED03 +
            _drop_state.get("salt", 0) * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


def set_seed_salt(salt):
    """Per-process salt of every dropout seed drawn from now on (sequence_parallel: rank + 1 inside the context, 0 outside)."""
    prev = _drop_state.get("salt", 0)
    _drop_state["salt"] = int(salt)
    return prev


def dropout(x, p, seed, out=None):
    """out = dropout(x) with the counter-based mask of `seed` (call again on the gradient for the backward)."""
    N, D = x.shape
    if out is None:
        out = torch.empty((N, D), dtype=x.dtype, device=x.device)
    px, ldx = _mat(x)
    po, ldo = _mat(out)
    tok = _pb(f"dropout ({N}x{D})", 2 * N * D * _es(x))
    L.check(L.lib().smx_dropout(dt(x), px, ldx, po, ldo, N, D, p, seed, _epoch(), _stream()), "smx_dropout")
    _pe(tok)
    return out


def add_rowtable(y, table, R):
    N, D = y.shape
    py, ldy = _mat(y)
    L.check(L.lib().smx_add_rowtable(dt(y), py, ldy, _p(table), R, N, D, _stream()), "smx_add_rowtable")
    return y


def cast(src, dtype):
    """fp32 <-> compute dtype through the library's cast kernels (contiguous tensors)."""
    if src.dtype == dtype:
        return src
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    if src.dtype == torch.float32:
        L.check(L.lib().smx_cast_from_f32(_DT[dtype], _p(src), _p(dst), src.numel(), _stream()), "smx_cast_from_f32")
    elif dtype == torch.float32:
        L.check(L.lib().smx_cast_to_f32(_DT[src.dtype], _p(src), _p(dst), src.numel(), _stream()), "smx_cast_to_f32")
    else:
        return cast(cast(src, torch.float32), dtype)
    return dst


def adamw_step(param, grad, m, v, shadow, lr, b1, b2, eps, wd, step, grad_scale=1.0, gscale_dev=None):
    tok = _pb(f"adamw ({param.numel()} params)", (28 + (2 if shadow is not None else 0)) * param.numel())
    L.check(L.lib().smx_adamw_step(_p(param), _p(grad), _p(m), _p(v), _p(shadow), param.numel(), lr, b1, b2, eps, wd,
                                   step, grad_scale, _p(gscale_dev), _epoch() if step <= 0 else None, _stream()), "smx_adamw_step")
    _pe(tok)


def utt_meanstd(x2, lens, mean, std, B, T, mean_norm=True, std_norm=True, eps=1e-10):
    px, ldx = _mat(x2)
    L.check(L.lib().smx_utt_meanstd(dt(x2), px, ldx, _p(lens), _p(mean), _p(std), B, T, x2.shape[1], int(mean_norm),
                                    int(std_norm), eps, _stream()), "smx_utt_meanstd")


def stats_combine(cur_mean, cur_std, glob_mean, glob_std, weight):
    B, F = cur_mean.shape
    L.check(L.lib().smx_stats_combine(_p(cur_mean), _p(cur_std), B, F, _p(glob_mean), _p(glob_std), float(weight), _stream()),
            "smx_stats_combine")


def colnorm(x2, mean, std, stat_stride, out, B, T):
    px, ldx = _mat(x2)
    po, ldo = _mat(out)
    L.check(L.lib().smx_colnorm(dt(x2), px, ldx, _p(mean), _p(std), stat_stride, po, ldo, B, T, x2.shape[1], _stream()),
            "smx_colnorm")


def log_softmax_fwd(x):
    """log_softmax over the last dim of a (N, V) view."""
    N, V = x.shape
    y = torch.empty((N, V), dtype=x.dtype, device=x.device)
    px, ldx = _mat(x)
    L.check(L.lib().smx_log_softmax_fwd(dt(x), px, ldx, _p(y), V, N, V, _stream()), "smx_log_softmax_fwd")
    return y


def log_softmax_bwd(dy, y):
    N, V = y.shape
    dx = torch.empty((N, V), dtype=y.dtype, device=y.device)
    pdy, lddy = _mat(dy)
    py, ldy = _mat(y)
    L.check(L.lib().smx_log_softmax_bwd(dt(y), pdy, lddy, py, ldy, _p(dx), V, N, V, _stream()), "smx_log_softmax_bwd")
    return dx


def ctc_fwd(lp2, targets, in_len, tgt_len, B, T, blank):
    """lp2: (B*T, V) log-probabilities; targets int32 (B, Smax); lengths int32 (B).  -> (nll (B) fp32, workspace)."""
    V, Smax = lp2.shape[1], targets.shape[1]
    assert targets.dtype == torch.int32 and in_len.dtype == torch.int32 and tgt_len.dtype == torch.int32
    assert targets.is_contiguous()
    nll = torch.empty((B,), dtype=torch.float32, device=lp2.device)
    ws = torch.empty(max(L.lib().smx_ctc_workspace(B, T, Smax), 16), dtype=torch.uint8, device=lp2.device)  # kept for the bwd
    plp, ldlp = _mat(lp2)
    L.check(L.lib().smx_ctc_loss_fwd(dt(lp2), plp, ldlp, _p(targets), _p(in_len), _p(tgt_len), B, T, V, Smax, blank, _p(nll),
                                     _p(ws), _stream()), "smx_ctc_loss_fwd")
    return nll, ws


def ctc_bwd(lp2, targets, in_len, tgt_len, B, T, blank, nll, gscale, ws):
    V, Smax = lp2.shape[1], targets.shape[1]
    g = torch.empty((B * T, V), dtype=lp2.dtype, device=lp2.device)
    plp, ldlp = _mat(lp2)
    L.check(L.lib().smx_ctc_loss_bwd(dt(lp2), plp, ldlp, _p(targets), _p(in_len), _p(tgt_len), B, T, V, Smax, blank, _p(nll),
                                     _p(gscale), _p(g), V, _p(ws), _stream()), "smx_ctc_loss_bwd")
    return g


_CSGU_DROP_FUSE = True   # (round 4: A/B knob SMX_CSGU_DROP_FUSE removed)     # (read once, like the library's own knobs)
_STEP_COUNTER = None          # the training loop's device step counter (held HERE, in the Python host; libsmx has no such state)


def _epoch():
    return None if _STEP_COUNTER is None else ctypes.c_void_p(_STEP_COUNTER.data_ptr())


def set_step_counter(counter):
    """Use (or stop using, with None) a device step counter - an int64 tensor of one element - in every dropout seed and
    AdamW bias correction issued through this module.  The library takes the counter as an explicit argument of each
    call (include/smx.h: `epoch`, smx_epilogue.epoch, `step_dev`); this module-level variable is the host's training
    loop state, the place torch keeps its own default generator."""
    global _STEP_COUNTER
    if counter is not None:
        assert counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1
    _STEP_COUNTER = counter


def step_counter_add(counter, inc=1):
    L.check(L.lib().smx_step_counter_add(_p(counter), inc, _stream()), "smx_step_counter_add")


def capture_id():
    """Id of the hipGraph capture the current stream records into; 0 when it is not capturing (smx_stream_capture_id)."""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    cid = ctypes.c_uint64(0)
    L.check(L.lib().smx_stream_capture_id(_stream(), ctypes.byref(cid)), "smx_stream_capture_id")
    return int(cid.value) or -1


def sumsq(x, out):
    ws = _workspace(L.lib().smx_sumsq_workspace(), x.device, slot=6)
    L.check(L.lib().smx_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), "smx_sumsq")


def clip_factor(sumsq_t, max_norm, inv_scale, out):
    L.check(L.lib().smx_clip_factor(_p(sumsq_t), max_norm, inv_scale, _p(out), _stream()), "smx_clip_factor")


# ---- front-end -----------------------------------------------------------------------------------------
def frame_window(wav, window, T, n_fft, hop):
    B, Lw = wav.shape
    out = torch.empty((B * T, n_fft), dtype=torch.float32, device=wav.device)
    L.check(L.lib().smx_frame_window(_p(wav), wav.stride(0), _p(window), _p(out), B, Lw, T, n_fft, hop, _stream()),
            "smx_frame_window")
    return out


def mel_db(spec, im_off, fb, B, T, amin, top_db, out_dtype):
    n_mels, n_bins = fb.shape
    out = torch.empty((B, T, n_mels), dtype=out_dtype, device=spec.device)
    ws = _workspace(L.lib().smx_fbank_workspace(B, T, n_mels), spec.device, slot=5)
    L.check(L.lib().smx_mel_db(_DT[out_dtype], _p(spec), spec.stride(0), im_off, _p(fb), n_bins, n_mels, amin, top_db, _p(out),
                               B, T, _p(ws), _stream()), "smx_mel_db")
    return out


def im2col_s2(x, Kp):
    B, T, F_, C = x.shape
    T2, F2 = (T + 1) // 2, (F_ + 1) // 2
    col = torch.empty((B * T2 * F2, Kp), dtype=x.dtype, device=x.device)
    L.check(L.lib().smx_im2col_s2(dt(x), _p(x), _p(col), B, T, F_, C, Kp, _stream()), "smx_im2col_s2")
    return col


def dft_frames(wav_padded, basis_cos, basis_sin, spec, im_off, B, T, n_fft, hop):
    """spec (B*T, lds) <- DFT of the overlapping frames of the zero-padded waveform rows (smx_dft_frames: no frame matrix, folded
    cosine / sine halves)."""
    tok = _pb(f"dft_frames ({B},{T}) n_fft={n_fft}", wav_padded.numel() * 4 + spec.numel() * 4, 2.0 * B * T * n_fft * (n_fft / 2 + 1))
    L.check(L.lib().smx_dft_frames(_p(wav_padded), wav_padded.stride(0), _p(basis_cos), _p(basis_sin), _p(spec), spec.stride(0), im_off,
                                   B, T, n_fft, hop, basis_cos.shape[0], _stream()), "smx_dft_frames")
    _pe(tok)


def conv1_ln_ok(x, O):
    """Does the fused first conv block (smx_conv1_ln_fwd / _bwd) take this input?  x (B, T, F) contiguous, one input channel."""
    B, T, F_ = x.shape
    return (x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous() and O == 64 and F_ % 16 == 0 and 16 <= F_ <= 160
            and T >= 2 and L.lib().smx_conv1_ln_workspace(B, T, F_, O) > 0)


def conv1_ln_fwd(x, w9, bias, gamma, beta, eps, act, need_stats=True):
    """a = act(LayerNorm(conv3x3_s2_reflect(x) + bias) * gamma + beta) in one pass: x (B,T,F) -> (B*T2, (F/2)*O), stats."""
    B, T, F_ = x.shape
    O = w9.shape[0]
    T2 = (T + 1) // 2
    y = torch.empty((B * T2, (F_ // 2) * O), dtype=x.dtype, device=x.device)
    stats = torch.empty((B * T2, 2), dtype=torch.float32, device=x.device) if need_stats else None
    tok = _pb(f"conv1_ln_fwd ({B},{T},{F_})->{O}", x.numel() * _es(x) + y.numel() * _es(y))
    L.check(L.lib().smx_conv1_ln_fwd(dt(x), _p(x), _p(w9), _p(bias), _p(gamma), _p(beta), eps, act, _p(y), _p(stats), B, T, F_, O,
                                     _stream()), "smx_conv1_ln_fwd")
    _pe(tok)
    return y, stats


def conv1_ln_bwd(da, x, w9, bias, gamma, beta, stats, act):
    """Gradients of the fused first conv block from dA and x: a flat fp32 tensor [dgamma | dbeta | dW9 (O,9) | dbias]."""
    B, T, F_ = x.shape
    O = w9.shape[0]
    D = (F_ // 2) * O
    grads = torch.zeros(2 * D + O * 10, dtype=torch.float32, device=x.device)
    ws = _workspace(L.lib().smx_conv1_ln_workspace(B, T, F_, O), x.device, slot=7)
    tok = _pb(f"conv1_ln_bwd ({B},{T},{F_})->{O}", da.numel() * _es(da) + x.numel() * _es(x))
    L.check(L.lib().smx_conv1_ln_bwd(dt(x), _p(da), _p(x), _p(w9), _p(bias), _p(gamma), _p(beta), _p(stats), act, _p(grads), _p(ws),
                                     B, T, F_, O, _stream()), "smx_conv1_ln_bwd")
    _pe(tok)
    return grads


def linear_k16(x, W, bias):
    """y = x W^T + bias for K = 16 on the VALU (smx_linear_k16_fwd), or None when the shape is not the kernel's."""
    N, K = x.shape
    M = W.shape[0]
    if not (x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and K == 16 and x.is_contiguous() and W.is_contiguous()
            and M % 8 == 0 and 256 % (M // 8) == 0):
        return None
    y = torch.empty((N, M), dtype=x.dtype, device=x.device)
    tok = _pb(f"linear_k16 ({N}x16)x(16x{M})", N * (K + M) * 2)
    L.check(L.lib().smx_linear_k16_fwd(dt(x), _p(x), _p(W), _p(bias), _p(y), N, M, _stream()), "smx_linear_k16_fwd")
    _pe(tok)
    return y


def conv2d_s2_direct_ok(x, O):
    """Do the patch-matrix-free kernels (smx_conv2d_s2_fwd / _wgrad) take this block input x (B, T, F, C)?"""
    B, T, F_, C = x.shape
    return (x.dtype == torch.bfloat16 and x.is_contiguous() and C == 64 and O % 8 == 0 and T >= 2 and F_ >= 2
            and x.numel() * 2 < (1 << 31) and x.data_ptr() % 16 == 0)


def conv2d_s2_fwd(x, wg, bias, O):
    """y (B*T2*F2, O) = conv3x3_s2_reflect(x) + bias with the GEMM gathering its operand from x (B,T,F,64) in place."""
    B, T, F_, C = x.shape
    rows = B * ((T + 1) // 2) * ((F_ + 1) // 2)
    y = torch.empty((rows, O), dtype=x.dtype, device=x.device)
    tok = _pb(f"conv2d_s2_fwd ({B},{T},{F_},{C})->{O}", x.numel() * 2 + y.numel() * 2, 2.0 * rows * O * 9 * C)
    L.check(L.lib().smx_conv2d_s2_fwd(dt(x), _p(x), _p(wg), _p(bias), _p(y), B, T, F_, C, O, wg.shape[1], _stream()), "smx_conv2d_s2_fwd")
    _pe(tok)
    return y


def conv2d_s2_wgrad(dy, x, gw, gbias):
    """gw (O, Kp) fp32 += dy^T patches(x), gbias (O) += column sums of dy (slab split-K, fixed-order reduction)."""
    B, T, F_, C = x.shape
    O = dy.shape[1]
    ws = _workspace(L.lib().smx_conv2d_s2_wgrad_workspace(B, T, F_, C, O), x.device, slot=6)
    tok = _pb(f"conv2d_s2_wgrad ({B},{T},{F_},{C})->{O}", x.numel() * 2 + dy.numel() * 2, 2.0 * dy.shape[0] * O * 9 * C)
    L.check(L.lib().smx_conv2d_s2_wgrad(dt(x), _p(dy), _p(x), _p(gw), _p(gbias), B, T, F_, C, O, gw.shape[1], _p(ws), _stream()),
            "smx_conv2d_s2_wgrad")
    _pe(tok)


def conv2d_s2_dgrad_ok(dy, C, O, T, F_):
    """Does the direct dgrad kernel (smx_conv2d_s2_dgrad) take this block?  (bf16, 64 -> 32 channels: the recipe's second)"""
    return dy.dtype == torch.bfloat16 and C == 64 and O == 32 and T >= 4 and F_ >= 4 and dy.numel() * 2 < (1 << 31)


def conv2d_s2_dgrad(dy, wg, B, T, F_, C):
    """dx (B,T,F,C) of the 3x3 / stride 2 / reflect-pad-1 convolution from dy (B*T2*F2, O) and the GEMM-layout weight."""
    dx = torch.empty((B, T, F_, C), dtype=dy.dtype, device=dy.device)
    tok = _pb(f"conv2d_s2_dgrad ({B},{T},{F_},{C})", dx.numel() * 2 + dy.numel() * 2)
    L.check(L.lib().smx_conv2d_s2_dgrad(dt(dy), _p(dy), _p(wg), _p(dx), B, T, F_, C, dy.shape[1], wg.shape[1], _stream()),
            "smx_conv2d_s2_dgrad")
    _pe(tok)
    return dx


def col2im_s2(dcol, B, T, F_, C):
    dx = torch.empty((B, T, F_, C), dtype=dcol.dtype, device=dcol.device)
    L.check(L.lib().smx_col2im_s2(dt(dcol), _p(dcol), _p(dx), B, T, F_, C, dcol.shape[1], _stream()), "smx_col2im_s2")
    return dx
