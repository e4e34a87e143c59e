"""Kernel-level GPU tests through the C-ABI: every GEMM layout / tile / alignment path against a plain
float64 matmul, epilogue fields, LayerNorm, depthwise conv, masked mean, chunk mean, AdamW."""
import math

import pytest
import torch

from tests._util import rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from summarymixing_amd import _lib as L, ops
    return L, ops


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("N,M,K", [(300, 200, 136), (64, 64, 64), (1000, 512, 256), (37, 19, 23), (129, 257, 70),
                                   (4000, 1024, 512)])
def test_gemm_layouts(N, M, K, dtype, tol):
    L, ops = _ops()
    torch.manual_seed(N + M + K)
    a = torch.randn(N, K, device="cuda").to(dtype)
    # asymmetric operands (detects transposed fragments / outputs)
    b_nt = (torch.randn(M, K, device="cuda") + torch.linspace(-1, 1, K, device="cuda")[None]).to(dtype)
    ref = a.double() @ b_nt.double().t()
    c = torch.empty(N, M, device="cuda", dtype=dtype)
    ops.gemm(L.GEMM_NT, a, b_nt, c, N, M, K)
    assert rel_err(c, ref) <= tol
    b_nn = b_nt.t().contiguous()                       # (K, M)
    c2 = torch.empty(N, M, device="cuda", dtype=dtype)
    ops.gemm(L.GEMM_NN, a, b_nn, c2, N, M, K)
    assert rel_err(c2, ref) <= tol
    a_t = a.t().contiguous()                           # (K, N)
    c3 = torch.zeros(N, M, device="cuda", dtype=torch.float32)
    ops.gemm(L.GEMM_TN, a_t, b_nn, c3, N, M, K, ops.epilogue(out_mode=L.OUT_ATOMIC_F32), splits=3)
    assert rel_err(c3, ref) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_gemm_epilogue_all_fields(dtype, tol):
    L, ops = _ops()
    torch.manual_seed(3)
    N, M, K, T = 96 * 5, 160, 72, 96
    x = torch.randn(N, K, device="cuda").to(dtype)
    w = torch.randn(M, K, device="cuda").to(dtype) * 0.2
    bias = torch.randn(M, device="cuda")
    c0 = torch.randn(N // T, M, device="cuda")
    res = torch.randn(N, M, device="cuda").to(dtype)
    mask = (torch.rand(N, device="cuda") > 0.3)
    z = torch.empty(N, M, device="cuda", dtype=dtype)
    y = torch.empty(N, M, device="cuda", dtype=dtype)
    for act, fn in [(L.ACT_GELU, torch.nn.functional.gelu), (L.ACT_SWISH, torch.nn.functional.silu),
                    (L.ACT_LEAKY_RELU, torch.nn.functional.leaky_relu), (L.ACT_RELU, torch.relu)]:
        e = ops.epilogue(bias=bias, c0=c0, c0_mode=L.C0_GROUP, c0_div=T, act=act, z=z, row_mask=mask.view(torch.uint8),
                         res=res, alpha=0.5)
        ops.gemm(L.GEMM_NT, x, w, y, N, M, K, e)
        v = x.double() @ w.double().t() + bias.double() + c0.double().repeat_interleave(T, 0)
        ref = res.double() + 0.5 * fn(v) * mask.double()[:, None]
        assert rel_err(z, v) <= tol
        assert rel_err(y, ref) <= tol
    pe = torch.randn(T, M, device="cuda")
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(c0=pe, c0_mode=L.C0_MOD, c0_div=T))
    assert rel_err(y, x.double() @ w.double().t() + pe.double().repeat(N // T, 1)) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_gemm_strided_views_and_batch(dtype, tol):
    """Column slices via leading dimensions + batched heads (the ParallelLinear einsum)."""
    L, ops = _ops()
    torch.manual_seed(4)
    N, H, f, h = 333, 4, 24, 40
    x = torch.randn(N, H * f, device="cuda").to(dtype)
    w = torch.randn(H, f, h, device="cuda").to(dtype)
    b = torch.randn(H, h, device="cuda")
    y = torch.empty(N, H * h, device="cuda", dtype=dtype)
    ops.gemm(L.GEMM_NN, x[:, :f], w[0], y[:, :h], N, h, f, ops.epilogue(bias=b, bias_batch_stride=h), batch=H, sa=f,
             sb=f * h, sc=h, lda=H * f, ldb=h, ldc=H * h)
    ref = torch.einsum("nmf,mfh->nmh", x.double().view(N, H, f), w.double()) + b.double()
    assert rel_err(y, ref.reshape(N, H * h)) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("D", [256, 144, 30, 512, 2560, 3332, 4096, 2050,     # > 2048: the wide-row kernels (front-end: 40 x 64)
                               1536, 1032, 2048])                              # 1024 < D <= 2048: workgroup-per-row kernels (bf16)
def test_layernorm_fwd_bwd(D, dtype, tol):
    L, ops = _ops()
    torch.manual_seed(D)
    N = 777
    x = (torch.randn(N, D, device="cuda") * 2 + 0.5).to(dtype)
    g = torch.randn(D, device="cuda") * 0.3 + 1
    b = torch.randn(D, device="cuda") * 0.3
    for act, fn in [(L.ACT_NONE, lambda v: v), (L.ACT_SWISH, torch.nn.functional.silu),
                    (L.ACT_LEAKY_RELU, lambda v: torch.nn.functional.leaky_relu(v, 0.01))]:
        y, stats = ops.layernorm_fwd(x, g, b, 1e-5, True, act)
        xr = x.double().requires_grad_(True)
        gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
        ref = fn(torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5))
        assert rel_err(y, ref) <= tol
        dy = torch.randn(N, D, device="cuda").to(dtype)
        res = torch.randn(N, D, device="cuda").to(dtype)
        (ref * dy.double()).sum().backward()
        dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        dx = ops.layernorm_bwd(dy, x, g, b, stats, dg, db, res, act)
        assert rel_err(dx, xr.grad + res.double()) <= 2 * tol
        assert rel_err(dg, gr.grad) <= 3 * tol and rel_err(db, br.grad) <= 3 * tol


@pytest.mark.gpu
@pytest.mark.parametrize("D,ld", [(1536, 3072), (512, 512), (2048, 2056), (40, 48)])
@pytest.mark.parametrize("zact", ["gelu", "swish"])
def test_layernorm_backward_through_the_producing_activation(D, ld, zact):
    """smx_layernorm_bwd_preact: x = act(z) feeds a LayerNorm; the kernel returns dL/dz = act'(z) * LNbwd(dy) (strided views:
    the gate half of the cgMLP's (N, 3072) tensors), dgamma / dbeta as the plain backward."""
    L, ops = _ops()
    torch.manual_seed(D + ld)
    N = 1203
    bf = torch.bfloat16
    code, fn = (L.ACT_GELU, torch.nn.functional.gelu) if zact == "gelu" else (L.ACT_SWISH, torch.nn.functional.silu)
    zfull = torch.randn(N, ld, device="cuda").to(bf)
    z = zfull[:, ld - D:]
    xfull = fn(zfull.float()).to(bf)
    x = xfull[:, ld - D:]
    g = torch.randn(D, device="cuda") * 0.3 + 1
    b = torch.randn(D, device="cuda") * 0.3
    y, stats = ops.layernorm_fwd(x, g, b, 1e-5, True)
    dy = torch.randn(N, D, device="cuda").to(bf)
    zr = z.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    xr = fn(zr)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    assert rel_err(y, ref) <= 2e-2
    (ref * dy.double()).sum().backward()
    out = torch.zeros(N, ld, device="cuda", dtype=bf)
    dz_view = out[:, ld - D:]
    assert ops.layernorm_bwd_preact_ok(dy, x, z, dz_view)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dz = ops.layernorm_bwd_preact(dy, x, g, b, stats, z, code, dg, db, dx_out=dz_view)
    assert rel_err(dz, zr.grad) <= 3e-2                  # (x is the bf16 rounding of act(z): the LN statistics see that rounding)
    assert rel_err(dg, gr.grad) <= 3e-2 and rel_err(db, br.grad) <= 3e-2
    if ld > D:
        assert float(out[:, :ld - D].abs().max()) == 0.0     # nothing outside the view was touched
    # the two-pass path it replaces gives the same values up to one bf16 rounding of the intermediate gradient
    dx = ops.layernorm_bwd(dy, x, g, b, stats, torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"))
    two = ops.act_mask_bwd(dx, z, None, code, 1.0, torch.empty_like(dx), None)
    assert rel_err(dz, two.double()) <= 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("D", [256, 512, 1536, 2048, 200])
def test_layernorm_of_the_float32_stream(D):
    """LayerNorm(float32 row) -> bf16 output, and its backward with bf16 gradients next to the float32 input
    (smx_layernorm_fwd_x32 / smx_layernorm_bwd2_x32: the fp32 residual stream of a bf16 model), every row-width class."""
    L, ops = _ops()
    torch.manual_seed(D)
    N = 1001
    bf = torch.bfloat16
    x = torch.randn(N, D, device="cuda") * 2 + 0.5
    g = torch.randn(D, device="cuda") * 0.3 + 1
    b = torch.randn(D, device="cuda") * 0.3
    y, stats = ops.layernorm_fwd(x, g, b, 1e-5, True, out_dtype=bf)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    assert y.dtype == bf and rel_err(y, ref) <= 1e-2
    dy = torch.randn(N, D, device="cuda").to(bf)
    res = torch.randn(N, D, device="cuda").to(bf)
    (ref * dy.double()).sum().backward()
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx = ops.layernorm_bwd(dy, x, g, b, stats, dg, db, res)
    assert dx.dtype == bf and rel_err(dx, xr.grad + res.double()) <= 2e-2
    assert rel_err(dg, gr.grad) <= 3e-2 and rel_err(db, br.grad) <= 3e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,T,D", [(3, 1000, 256), (2, 77, 40), (8, 3000, 512), (1, 5, 8)])
def test_masked_mean_and_broadcast(B, T, D, dtype, tol):
    L, ops = _ops()
    torch.manual_seed(T)
    buf = torch.randn(B * T, 2 * D, device="cuda").to(dtype)
    s = buf[:, D:]                                         # strided view (ld = 2D) like the fast-mode split
    lens = torch.randint(1, T + 1, (B,), device="cuda")
    lens[0] = T
    mask = (torch.arange(T, device="cuda")[None] < lens[:, None])
    out, inv = ops.masked_mean(s, mask.reshape(-1).view(torch.uint8), B, T, True, True)
    ref = (s.double().view(B, T, D) * mask[..., None]).sum(1) / lens[:, None]
    assert rel_err(out, ref) <= tol
    assert rel_err(inv, 1.0 / lens.double()) <= 1e-6
    out2, _ = ops.masked_mean(s, None, B, T, False)
    assert rel_err(out2, s.double().view(B, T, D).sum(1)) <= tol
    # bit-reproducible (fixed-order split-T combine)
    out3, _ = ops.masked_mean(s, mask.reshape(-1).view(torch.uint8), B, T, True)
    assert torch.equal(out, out3)
    ds = torch.empty(B * T, D, device="cuda", dtype=dtype)
    ops.bcast_rows(out, inv, ds, B, T)
    assert rel_err(ds.view(B, T, D), (ref / lens[:, None]).unsqueeze(1).expand(B, T, D)) <= tol


@pytest.mark.parametrize("left", [None, 0, 2])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_chunk_mean_fwd_bwd(left, dtype, tol):
    from summarymixing_amd.functional import DynChunkMask
    L, ops = _ops()
    torch.manual_seed(5)
    B, T, D, chunk = 3, 70, 48, 8
    s = torch.randn(B * T, D, device="cuda").to(dtype)
    Mx = DynChunkMask(T, chunk, left).dense("cuda").double()
    out = torch.empty_like(s)
    ops.chunk_mean(s, out, B, T, chunk, left)
    ref = (Mx @ s.double().view(B, T, D)) / Mx.sum(1)[None, :, None]
    assert rel_err(out.view(B, T, D), ref) <= tol
    g = torch.randn(B * T, D, device="cuda").to(dtype)
    ds = torch.empty_like(s)
    ops.chunk_mean(g, ds, B, T, chunk, left, reverse=True)
    refb = (Mx / Mx.sum(1)[:, None]).t() @ g.double().view(B, T, D)
    assert rel_err(ds.view(B, T, D), refb) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("B,T,D,k,chunk", [(2, 150, 96, 31, 0), (3, 70, 40, 7, 0), (2, 100, 64, 31, 16), (1, 9, 8, 5, 4),
                                           # D % 64 == 0, k = 31, no chunking: the rolling register-window kernels (dwconv_roll.h)
                                           (2, 150, 128, 31, 0), (3, 500, 64, 31, 0), (1, 9, 64, 31, 0), (5, 131, 192, 31, 0),
                                           (128, 500, 256, 31, 0),
                                           # Dynamic Chunk Convolution in the rolling kernels: chunks that divide / straddle the
                                           # 16-frame steps and the 128-frame wave segments, chunk > T, chunk = 1
                                           (2, 150, 128, 31, 8), (1, 300, 64, 31, 13), (2, 200, 64, 31, 32), (1, 130, 64, 31, 50),
                                           (3, 257, 64, 31, 24), (2, 100, 64, 31, 7), (1, 40, 64, 31, 100), (2, 70, 64, 31, 1),
                                           (4, 500, 256, 31, 4), (4, 500, 256, 31, 16), (16, 500, 256, 31, 8)])
def test_glu_dwconv_fwd_bwd(B, T, D, k, chunk, dtype, tol):
    from oracle import smx_oracle as O
    L, ops = _ops()
    torch.manual_seed(T + k)
    p = torch.randn(B * T, 2 * D, device="cuda").to(dtype)
    w = torch.randn(D, k, device="cuda") * 0.3
    bias = torch.randn(D, device="cuda")
    y = ops.dwconv_fwd(p, w, bias, B, T, D, k, True, L.PAD_ZERO, chunk)
    pr = p.double().cpu().view(B, T, 2 * D).requires_grad_(True)
    wr, br = w.double().cpu().requires_grad_(True), bias.double().cpu().requires_grad_(True)
    u = pr[..., :D] * torch.sigmoid(pr[..., D:])
    if chunk:
        ref = O.depthwise_conv_chunked(u, wr.view(D, 1, k), br, chunk)
    else:
        ref = torch.nn.functional.conv1d(u.transpose(1, 2), wr.view(D, 1, k), br, padding=(k - 1) // 2, groups=D).transpose(1, 2)
    assert rel_err(y.view(B, T, D), ref) <= tol
    dy = torch.randn(B * T, D, device="cuda").to(dtype)
    (ref * dy.double().cpu().view(B, T, D)).sum().backward()
    dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
    dp, _ = ops.dwconv_bwd(dy, p, w, bias, dw, db, B, T, D, k, True, L.PAD_ZERO, chunk)
    assert rel_err(dp.view(B, T, 2 * D), pr.grad) <= 2 * tol
    assert rel_err(dw, wr.grad) <= 2 * tol and rel_err(db, br.grad) <= 2 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("B,T,D,k", [(2, 90, 48, 7), (2, 150, 64, 31), (1, 40, 16, 31), (3, 129, 72, 31),
                                     # D % 64 == 0, k = 31, bf16: the rolling CSGU kernels + edge-fold kernel (dwconv_roll.h)
                                     (4, 250, 128, 31), (2, 31, 64, 31), (3, 16, 64, 31), (2, 47, 192, 31), (16, 250, 1536, 31)])
def test_gated_reflect_dwconv_fwd_bwd(B, T, D, k, dtype, tol):
    """Branchformer CSGU form: y = gate * conv_reflect(x)  (k=31 takes the register-window fast path)."""
    L, ops = _ops()
    torch.manual_seed(11 + T)
    x = torch.randn(B * T, D, device="cuda").to(dtype)
    gate = torch.randn(B * T, D, device="cuda").to(dtype)
    w = torch.randn(D, k, device="cuda") * 0.3
    bias = torch.randn(D, device="cuda")
    y = ops.dwconv_fwd(x, w, bias, B, T, D, k, False, L.PAD_REFLECT, 0, gate)
    xr = x.double().cpu().view(B, T, D).requires_grad_(True)
    gr = gate.double().cpu().view(B, T, D).requires_grad_(True)
    wr, br = w.double().cpu().requires_grad_(True), bias.double().cpu().requires_grad_(True)
    xp = torch.nn.functional.pad(xr.transpose(1, 2), (k // 2, k // 2), mode="reflect")
    ref = torch.nn.functional.conv1d(xp, wr.view(D, 1, k), br, groups=D).transpose(1, 2) * gr
    assert rel_err(y.view(B, T, D), ref) <= tol
    dy = torch.randn(B * T, D, device="cuda").to(dtype)
    (ref * dy.double().cpu().view(B, T, D)).sum().backward()
    dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
    dx, dgate = ops.dwconv_bwd(dy, x, w, bias, dw, db, B, T, D, k, False, L.PAD_REFLECT, 0, gate)
    assert rel_err(dx.view(B, T, D), xr.grad) <= 2 * tol
    assert rel_err(dgate.view(B, T, D), gr.grad) <= 2 * tol
    assert rel_err(dw, wr.grad) <= 2 * tol and rel_err(db, br.grad) <= 2 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("N,M,K", [(700, 256, 136), (333, 72, 40), (1500, 1024, 256), (130, 67, 24)])
def test_gemm_act_grad_epilogue(N, M, K, dtype, tol):
    """SMX_EPI_ACT_GRAD + colsum: the dgrad GEMM emits the UPSTREAM layer's dZ = alpha * D(dX * act'(Z)) * mask and
    its bias gradient in one launch; must equal the unfused act_mask_bwd path (same dropout seed)."""
    L, ops = _ops()
    torch.manual_seed(N + K)
    dz2 = torch.randn(N, K, device="cuda").to(dtype)
    w = (torch.randn(K, M, device="cuda") * 0.2).to(dtype)            # NN: dX (N,M) = dZ2 (N,K) W (K,M)
    z = torch.randn(N, M, device="cuda").to(dtype)
    mask = (torch.rand(N, device="cuda") > 0.3).view(torch.uint8)
    for act, drop, mk, alpha in [(L.ACT_SWISH, None, None, 1.0), (L.ACT_GELU, (0.2, 1234567), mask, 0.5),
                                 (L.ACT_RELU, (0.1, 99), None, 1.0), (L.ACT_NONE, None, mask, 1.0)]:
        out = torch.empty(N, M, device="cuda", dtype=dtype)
        gb = torch.full((M,), 0.25, device="cuda")
        e = ops.epilogue(act=act, act_grad_z=z, drop=drop, row_mask=mk, alpha=alpha, colsum=gb)
        ops.gemm(L.GEMM_NN, dz2, w, out, N, M, K, e)
        dx = (dz2.double() @ w.double()).float()                       # unfused reference on the fp32 product
        ref = torch.empty(N, M, device="cuda")
        gb_ref = torch.full((M,), 0.25, device="cuda")
        ops.act_mask_bwd(dx, z.float(), mk, act, alpha, ref, gb_ref, drop=drop)
        assert rel_err(out, ref) <= tol
        assert rel_err(gb, gb_ref) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("rows,M,K,batch", [(3000, 256, 128, 1), (777, 72, 40, 1), (5000, 1024, 256, 1), (900, 48, 32, 3)])
def test_wgrad_with_bias_gradient(rows, M, K, batch, dtype, tol):
    """smx_linear_wgrad(dbias=): the column sums of dZ (the bias gradient) come out of the wgrad launch itself."""
    L, ops = _ops()
    torch.manual_seed(rows + M)
    dz = torch.randn(rows, batch * M, device="cuda").to(dtype)
    x = torch.randn(rows, batch * K, device="cuda").to(dtype)
    gW = torch.full((batch, M, K), 0.5, device="cuda")
    gb = torch.full((batch, M), -0.25, device="cuda")
    ops.wgrad(dz[:, :M], x[:, :K], gW[0], rows, M, K, batch=batch, sz=M, sx=K, sw=M * K, lddz=batch * M, ldx=batch * K,
              lddw=K, alpha=0.5, dbias=gb)
    dz3, x3 = dz.double().view(rows, batch, M), x.double().view(rows, batch, K)
    assert rel_err(gW, 0.5 + 0.5 * torch.einsum("rbm,rbk->bmk", dz3, x3)) <= tol
    assert rel_err(gb, -0.25 + 0.5 * dz3.sum(0)) <= tol


def test_act_mask_bwd_reductions():
    L, ops = _ops()
    torch.manual_seed(6)
    N, M, T = 5 * 64 + 7 * 0, 136, 64
    dy = torch.randn(N, M, device="cuda")
    z = torch.randn(N, M, device="cuda")
    mask = torch.rand(N, device="cuda") > 0.4
    dz = torch.empty_like(dy)
    db = torch.zeros(M, device="cuda")
    dg = torch.zeros(N // T, M, device="cuda")
    ops.act_mask_bwd(dy, z, mask.view(torch.uint8), L.ACT_GELU, 0.5, dz, db, dg, T)
    zr = z.double().requires_grad_(True)
    (torch.nn.functional.gelu(zr) * mask.double()[:, None] * 0.5 * dy.double()).sum().backward()
    assert rel_err(dz, zr.grad) <= 1e-5
    assert rel_err(db, zr.grad.sum(0)) <= 1e-5
    assert rel_err(dg, zr.grad.view(N // T, T, M).sum(1)) <= 1e-5


def test_adamw_matches_torch():
    L, ops = _ops()
    torch.manual_seed(7)
    n = 10007
    p0 = torch.randn(n, device="cuda")
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=8e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 4):
        g = torch.randn(n, device="cuda")
        ref_p.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g, m, v, sh, 8e-4, 0.9, 0.98, 1e-9, 0.01, step)
    assert rel_err(p, ref_p) <= 1e-6
    assert torch.equal(sh, p.to(torch.bfloat16))
    ss = torch.zeros(1, device="cuda")
    ops.sumsq(p, ss)
    assert abs(ss.item() - float((p.double() ** 2).sum())) / ss.item() < 1e-5
    cf = torch.zeros(1, device="cuda")
    ops.clip_factor(ss, 5.0, 1.0, cf)
    assert abs(cf.item() - min(1.0, 5.0 / (math.sqrt(ss.item()) + 1e-6))) < 1e-6


def test_reduce_jobs_batched():
    """smx_reduce_jobs: several fixed-order reductions (vector / scalar paths, strided destinations, few / many sources)
    in one launch, with a cached job table."""
    import ctypes
    L, ops = _ops()
    from summarymixing_amd import functional as F
    torch.manual_seed(9)
    cases = [(40, 64, 128, 128), (3, 5, 31, 40), (130, 1, 256, 256), (17, 8, 12, 20)]      # nsrc, rows, cols, ldd
    srcs, dsts, refs = [], [], []
    F._Deferred.jobs, F._Deferred.pending = [], set()
    for nsrc, rows, cols, ldd in cases:
        src = torch.randn(nsrc, rows, cols, device="cuda")
        dst = torch.randn(rows, ldd, device="cuda")
        refs.append(dst[:, :cols].double() + 0.5 * src.double().sum(0))
        F._Deferred.ws[len(srcs)] = src.view(-1).view(torch.uint8)          # (flush looks up the device there)
        F.defer(src.data_ptr(), dst[:, :cols], rows * cols, nsrc, rows, cols, alpha=0.5)
        srcs.append(src); dsts.append(dst)
    F.flush_deferred()
    for (nsrc, rows, cols, ldd), dst, ref in zip(cases, dsts, refs):
        assert rel_err(dst[:, :cols], ref) <= 1e-6
    assert not F._Deferred.jobs


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,T,D", [(2, 50, 64), (3, 333, 256), (1, 2000, 32), (2, 16, 8)])
def test_expdecay_mean_matches_dense_laplace(B, T, D, dtype, tol):
    """O(T) two-sided exponential filter == the reference's dense (T,T) Laplace path (summary_mixing.py:316-365,233-235),
    forward and transposed (backward) operator."""
    L, ops = _ops()
    torch.manual_seed(T + D)
    decay = 0.995 if T > 100 else 0.9
    s = torch.randn(B * T, D, device="cuda").to(dtype)
    idx = torch.arange(T, device="cuda")
    M = torch.pow(torch.tensor(decay, dtype=torch.float64, device="cuda"), (idx[None] - idx[:, None]).abs().double())
    Wn = M / M.sum(1, keepdim=True)
    s3 = s.double().view(B, T, D)
    out = torch.empty_like(s)
    ops.expdecay_mean(s, out, B, T, decay)
    assert rel_err(out.view(B, T, D), torch.einsum("ij,bjd->bid", Wn, s3)) <= tol
    ops.expdecay_mean(s, out, B, T, decay, reverse=True)
    assert rel_err(out.view(B, T, D), torch.einsum("ji,bjd->bid", Wn, s3)) <= tol


def test_gemm_large_ragged_shapes_and_epilogues():
    """The 128 x 128 tile at chip-filling sizes with ragged N / M / K: NT + bias + Swish + saved Z + row mask, NN with a
    residual and fp32 output, the fused activation-gradient epilogue with column sums - against fp32 torch references."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(0)
    for (N, K, M) in ((32768 + 77, 256, 1024), (32768, 192, 520), (33000, 64, 640)):
        x = torch.randn(N, K, device="cuda").bfloat16()
        w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(M, device="cuda")
        mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
        res = torch.randn(N, M, device="cuda").bfloat16()
        zr = x.float() @ w.float().t() + b

        def rel(a, r):
            return (a.float() - r).abs().max().item() / r.abs().max().item()
        y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
        z = torch.empty_like(y)
        ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, row_mask=mask))
        assert rel(z, zr) < 1e-2 and rel(y, torch.nn.functional.silu(zr) * mask[:, None]) < 1e-2, (N, K, M, "NT")
        wt = w.t().contiguous()
        y32 = torch.empty(N, M, device="cuda", dtype=torch.float32)
        ops.gemm(L.GEMM_NN, x, wt, y32, N, M, K, ops.epilogue(bias=b, res=res, out_mode=L.OUT_F32))
        assert rel(y32, zr + res.float()) < 1e-2, (N, K, M, "NN")
        zz = torch.randn(N, M, device="cuda").bfloat16()
        cs = torch.zeros(M, device="cuda")
        ops.gemm(L.GEMM_NN, x, wt, y, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=zz, colsum=cs))
        zf = zz.float()
        sg = torch.sigmoid(zf)
        ref = (x.float() @ wt.float()) * (sg * (1 + zf * (1 - sg)))
        assert rel(y, ref) < 1e-2, (N, K, M, "act-grad")
        assert (cs - y.float().sum(0)).abs().max().item() / y.float().sum(0).abs().max().item() < 2e-2, (N, K, M, "colsum")


@pytest.mark.parametrize("N", [25600 + 77, 51200])
def test_gemm_256x256_tile_long_reduction(N):
    """K = 2048 -> 512 (the FFN down-projection's input gradient at d_model = 512) takes the 256 x 256 tile by default when its
    epilogue has no element-wise side input and writes dtype T (one workgroup per CU, software-pipelined K loop,
    smx_config.t256): NN plain and NT + bias + row mask + dropout with a bf16 output are that path (the NT / B_KC variant of the
    pipeline and its epilogue on a 256-row tile); the float32-output NT call and the heavier epilogues of the same shape
    (residual, saved Z, activation gradient + column sums) take the 128 x 256 tile by default - the child process of
    test_gemm_256x256_tile_every_eligible_shape (SMX_T256=2) sends those to the big tile too.  All against fp32 torch
    references, ragged last row tile."""
    from summarymixing_amd import _lib as L, ops
    assert L.get_config()["t256"] >= 1
    torch.manual_seed(N)
    K, M = 2048, 512
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
    res = torch.randn(N, M, device="cuda").bfloat16()
    zr = x.float() @ w.float().t() + b

    def rel(a, r):
        return (a.float() - r).abs().max().item() / r.abs().max().item()
    y = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    z = torch.empty_like(y)
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, row_mask=mask, res=res, alpha=0.5))
    assert rel(z, zr) < 1e-2
    assert rel(y, res.float() + 0.5 * torch.nn.functional.silu(zr) * mask[:, None]) < 1e-2
    y32 = torch.empty(N, M, device="cuda", dtype=torch.float32)
    ops.gemm(L.GEMM_NT, x, w, y32, N, M, K, ops.epilogue(bias=b, out_mode=L.OUT_F32))
    assert rel(y32, zr) < 1e-2
    # the 256 x 256 tile's NT variant: bias (+ row mask, + dropout), bf16 output, no element-wise side input
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, row_mask=mask))
    assert rel(y, zr * mask[:, None]) < 1e-2
    yd = torch.empty_like(y)
    ops.gemm(L.GEMM_NT, x, w, yd, N, M, K, ops.epilogue(bias=b, row_mask=mask, drop=(0.25, 4242)))
    kept = yd != 0
    live = (mask[:, None] != 0) & (zr.abs() > 1e-3)
    rate = 1.0 - float((kept & live).sum()) / float(live.sum())
    assert abs(rate - 0.25) < 5e-3, rate                   # the counter-based mask at its rate ...
    assert rel(torch.where(kept, yd.float() * 0.75, zr * mask[:, None]), zr * mask[:, None]) < 1e-2   # ... survivors scaled by 1 / (1 - p)
    yd2 = torch.empty_like(y)
    ops.gemm(L.GEMM_NT, x, w, yd2, N, M, K, ops.epilogue(bias=b, row_mask=mask, drop=(0.25, 4242)))
    assert torch.equal(yd, yd2)                            # ... and a pure function of (seed, element)
    wt = w.t().contiguous()                                # NN: (N, K) x (K, M)
    ops.gemm(L.GEMM_NN, x, wt, y, N, M, K)
    assert rel(y, x.float() @ wt.float()) < 1e-2
    zz = torch.randn(N, M, device="cuda").bfloat16()
    cs = torch.zeros(M, device="cuda")
    ops.gemm(L.GEMM_NN, x, wt, y, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=zz, colsum=cs))
    zf = zz.float()
    sg = torch.sigmoid(zf)
    assert rel(y, (x.float() @ wt.float()) * (sg * (1 + zf * (1 - sg)))) < 1e-2
    assert (cs - y.float().sum(0)).abs().max().item() / y.float().sum(0).abs().max().item() < 2e-2


@pytest.mark.parametrize("N,M,K", [(32768, 512, 256), (32768, 256, 1024), (16384, 1536, 512)])
def test_wgrad_lds_dma_kernel(N, M, K):
    """Shapes that take the LDS-DMA TN kernel (gemm_tn_dma_kernel: M, K multiples of 128, frames a multiple of 64, at least
    256 workgroups): dW += dZ^T X accumulates into a non-zero gradient, the bias gradient comes out of the extra
    all-ones MFMA; fp32 torch reference."""
    from summarymixing_amd import ops
    torch.manual_seed(N + M)
    dz = torch.randn(N, M, device="cuda").bfloat16()
    x = torch.randn(N, K, device="cuda").bfloat16()
    gw0 = torch.randn(M, K, device="cuda")
    gb0 = torch.randn(M, device="cuda")
    gw, gb = gw0.clone(), gb0.clone()
    ops.wgrad(dz, x, gw, N, M, K, dbias=gb)
    ref = gw0 + dz.float().t() @ x.float()
    refb = gb0 + dz.float().sum(0)
    assert (gw - ref).abs().max().item() / ref.abs().max().item() < 1e-4
    assert (gb - refb).abs().max().item() / refb.abs().max().item() < 1e-4
    # strided operands (column slices of wider buffers), no bias
    wide_z = torch.randn(N, M + 128, device="cuda").bfloat16()
    wide_x = torch.randn(N, K + 256, device="cuda").bfloat16()
    dzs, xs = wide_z[:, 128:], wide_x[:, 128:128 + K]
    gw2 = torch.zeros(M, K, device="cuda")
    ops.wgrad(dzs, xs, gw2, N, M, K, lddz=wide_z.stride(0), ldx=wide_x.stride(0))
    ref2 = dzs.float().t() @ xs.float()
    assert (gw2 - ref2).abs().max().item() / ref2.abs().max().item() < 1e-4


@pytest.mark.parametrize("N,K,M", [(40000 + 33, 256, 1024), (33000, 192, 520)])
def test_gemm_register_domain_epilogue(N, K, M):
    """Epilogues without a saved Z take the register-domain path of the 128 x 128 kernel (math on the accumulator fragments,
    bf16 staging of the finished tile); with a saved Z the fp32-staged path runs.  Both must agree with an fp32 torch
    reference, and - the dropout index is a function of (row, column) only - with each other bit for bit under dropout."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(N)
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
    zr = x.float() @ w.float().t() + b
    yr = torch.nn.functional.gelu(zr) * mask[:, None] * 0.5
    y1 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    ops.gemm(L.GEMM_NT, x, w, y1, N, M, K, ops.epilogue(bias=b, act=L.ACT_GELU, row_mask=mask, alpha=0.5))
    assert rel_err(y1.float(), yr) < 1e-2
    # NN layout (dgrad shaped), bias only, strided output (a column slice of a wider buffer)
    wide = torch.zeros(N, M + 64, device="cuda", dtype=torch.bfloat16)
    ops.gemm(L.GEMM_NN, x, w.t().contiguous(), wide[:, 64:], N, M, K, ops.epilogue(bias=b))
    assert rel_err(wide[:, 64:].float(), zr) < 1e-2 and float(wide[:, :64].abs().max()) == 0.0
    # dropout: register-domain path (no Z) vs fp32-staged path (Z requested) with the same seed
    ya, yb = torch.empty_like(y1), torch.empty_like(y1)
    z = torch.empty_like(y1)
    ops.gemm(L.GEMM_NT, x, w, ya, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, row_mask=mask, drop=(0.15, 4242)))
    ops.gemm(L.GEMM_NT, x, w, yb, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, row_mask=mask, drop=(0.15, 4242), z=z))
    assert torch.equal(ya == 0, yb == 0)
    assert rel_err(ya.float(), yb.float()) < 1e-6
    keep = (ya != 0).float().mean().item() / mask.float().mean().item()
    assert abs(keep - 0.85) < 0.01


# ---- grouped wgrad: all the weight gradients of a layer in one launch (smx_wgrad_group) ------------------------------
def _wgroup_case(rows, shapes, bias, seed=0, strided=False):
    """shapes: list of (M, K).  Returns max rel errors of dW / db against fp64 math on the bf16 inputs."""
    import ctypes
    from summarymixing_amd import _lib as L, functional as F, ops
    torch.manual_seed(seed)
    recs = []
    for i, (M, K) in enumerate(shapes):
        if strided:                                   # column slices of wider buffers (explicit leading dimensions)
            dz = (torch.randn(rows, M + 64, device="cuda") * 0.5).bfloat16()[:, 32:32 + M]
            x = torch.randn(rows, 2 * K, device="cuda").bfloat16()[:, K:]
        else:
            dz = (torch.randn(rows, M, device="cuda") * 0.5).bfloat16()
            x = torch.randn(rows, K, device="cuda").bfloat16()
        gW = torch.randn(M, K, device="cuda")          # accumulate semantics: += on top of what is there
        gb = torch.randn(M, device="cuda") if bias[i] else None
        recs.append((dz, x, gW, gb, gW.clone(), gb.clone() if gb is not None else None))
    for dz, x, gW, gb, _, _ in recs:
        F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
    assert len(F._Deferred.group) == len(shapes), "the grouped path must take these shapes"
    F.flush_deferred()
    torch.cuda.synchronize()
    errs = []
    for dz, x, gW, gb, gW0, gb0 in recs:
        ref = gW0.double() + dz.double().t() @ x.double()
        errs.append(float((gW.double() - ref).abs().max() / ref.abs().max()))
        if gb is not None:
            refb = gb0.double() + dz.double().sum(0)
            errs.append(float((gb.double() - refb).abs().max() / refb.abs().max()))
    return max(errs)


@pytest.mark.parametrize("rows", [2048, 4160, 64000, 33000, 3750, 2111, 2049, 2080,    # rows % 64 = 40, 38, 63, 1, 32: the ragged tail inside the kernel
                                  64, 100, 500, 1000])                                # one utterance: a single slice of 1-15 steps
def test_wgrad_group_conformer_layer_shapes(rows):
    shapes = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)]
    err = _wgroup_case(rows, shapes, [True] * 8)
    assert err < 2e-5, err


def test_wgrad_group_strided_operands_mixed_bias_and_single_item():
    assert _wgroup_case(8192, [(512, 512), (256, 768)], [False, True], seed=1, strided=True) < 2e-5
    assert _wgroup_case(16384, [(256, 256)], [True], seed=2) < 2e-5
    assert _wgroup_case(8192 + 45, [(512, 512), (256, 768)], [True, False], seed=4, strided=True) < 2e-5     # ragged tail, strided views
    # more weights than one launch takes (SMX_WGRAD_GROUP_MAX = 16): split into two launches
    assert _wgroup_case(4096, [(256, 256)] * 18, [True, False] * 9, seed=3) < 2e-5


def test_wgrad_group_is_bit_reproducible_and_matches_the_slab_path():
    from summarymixing_amd import functional as F, ops
    _wgroup_case(4096, [(256, 256)], [True], seed=5)      # a small block first (its workspace / job table must not leak into the next flush)
    torch.manual_seed(4)
    rows, M, K = 20000, 512, 256
    dz = (torch.randn(rows, M, device="cuda") * 0.5).bfloat16()
    x = torch.randn(rows, K, device="cuda").bfloat16()
    outs = []
    for _ in range(2):
        gW, gb = torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")
        F._wgrad(dz, x, gW, rows, M, K, gb)
        F.flush_deferred()                                 # (the contract: gradients are final after flush_deferred)
        outs.append((gW.clone(), gb.clone()))
    dW, db = (outs[0][0] - outs[1][0]).abs(), (outs[0][1] - outs[1][1]).abs()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (
        f"dW: {int((dW > 0).sum())} differ, max {float(dW.max())}, nan {int(torch.isnan(dW).sum())}, rows {(dW > 0).nonzero()[:, 0].unique()[:6].tolist()}, "
        f"cols {(dW > 0).nonzero()[:, 1].unique()[:6].tolist()}; db: {int((db > 0).sum())} differ, max {float(db.max())}")
    gW2, gb2 = torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")
    ops.wgrad(dz, x, gW2, rows, M, K, dbias=gb2)        # the per-weight slab GEMM + reduction
    assert rel_err(outs[0][0], gW2) < 1e-5 and rel_err(outs[0][1], gb2) < 1e-5


def test_wgrad_tied_weight_across_flushes():
    """The same gradient buffer (a weight shared by two blocks) fed by several small blocks, one flush each: the slab workspace is
    reused, the reductions accumulate."""
    from summarymixing_amd import functional as F
    torch.manual_seed(11)
    rows, M, K = 2048, 256, 256
    gW, gb = torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")
    ref, refb = torch.zeros(M, K, dtype=torch.float64, device="cuda"), torch.zeros(M, dtype=torch.float64, device="cuda")
    for _ in range(6):
        dz = (torch.randn(rows, M, device="cuda") * 0.5).bfloat16()
        x = torch.randn(rows, K, device="cuda").bfloat16()
        F._wgrad(dz, x, gW, rows, M, K, gb)
        F.flush_deferred()
        ref += dz.double().t() @ x.double()
        refb += dz.double().sum(0)
    assert rel_err(gW, ref) < 1e-5 and rel_err(gb, refb) < 1e-5


# ---- LayerNorm fused into the row-complete GEMM epilogues (SMX_EPI_LN_BWD / SMX_EPI_LN_FWD): 128 x 256 tile (d_model = 256, two
# workgroups per CU) and 128 x 512 tile (d_model = 512, one workgroup per CU on the software-pipelined main loop) ------------------
@pytest.mark.parametrize("D", [256, 512])
@pytest.mark.parametrize("N,K", [(4096, 1024), (33000 + 77, 512), (200, 256), (2000 + 13, 2048), (50000 + 5, 256)])
def test_gemm_epilogue_layernorm_backward(N, K, D):
    """dgrad GEMM whose epilogue runs the LayerNorm backward (rows complete in the tile): dX, the second output
    alpha * D(dX) * mask, and the per-tile dgamma / dbeta partial rows against fp64 torch math."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(N)
    dz = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(K, D, device="cuda") * 0.05).bfloat16()           # NN: dh = dz @ W
    x = torch.randn(N, D, device="cuda").bfloat16()
    res = torch.randn(N, D, device="cuda").bfloat16()
    gamma = torch.randn(D, device="cuda") * 0.5 + 1.0
    mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    stats = torch.cat([mean, rstd], 1).float().contiguous()
    tr = L.lib().smx_gemm_ln_tile_rows_for(N, D)      # (64-row tiles at d_model 256 up to ~38 000 frames: round 6)
    ntile = (N + tr - 1) // tr
    partial = torch.zeros(ntile, 2, D, device="cuda")
    dx, dx2 = torch.empty(N, D, device="cuda", dtype=torch.bfloat16), torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    e = ops.epilogue(res=res, ln_bwd=(x, stats, gamma, partial, dx2, (0.5, mask, None)))
    ops.gemm(L.GEMM_NN, dz, W, dx, N, D, K, e)
    g = dz.double() @ W.double()
    xh = (xd - mean) * rstd
    gg = g * gamma.double()
    ref = rstd * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)) + res.double()
    assert rel_err(dx, ref) < 1e-2
    assert rel_err(dx2, 0.5 * ref * mask.bool()[:, None]) < 1e-2
    assert rel_err(partial[:, 0].sum(0), (g * xh).sum(0)) < 2e-3 and rel_err(partial[:, 1].sum(0), g.sum(0)) < 2e-3
    # the second output with dropout draws the mask of smx_dropout(seed) on the (N, 256) index space
    seed = 0x1234567
    e = ops.epilogue(res=res, ln_bwd=(x, stats, gamma, partial, dx2, (1.0, None, (0.25, seed))))
    ops.gemm(L.GEMM_NN, dz, W, dx, N, D, K, e)
    keep = ops.dropout(torch.ones(N, D, device="cuda"), 0.25, seed) != 0
    assert torch.equal(dx2 != 0, keep & (dx2 != 0)) and ((dx2 != 0) | ~keep | (dx.float().abs() < 1e-3)).all()
    assert rel_err(dx2, torch.where(keep, ref / 0.75, torch.zeros_like(ref))) < 1e-2


@pytest.mark.parametrize("D", [256, 512])
@pytest.mark.parametrize("N,K,act", [(4096, 1024, 0), (33000 + 77, 512, 2), (200, 256, 0), (2000 + 13, 2048, 0)])
def test_gemm_epilogue_layernorm_forward(N, K, act, D):
    """NT GEMM + bias + residual whose epilogue appends y = act(LN(C)) and the row statistics."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(N + 1)
    a = torch.randn(N, K, device="cuda").bfloat16()
    W = (torch.randn(D, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(D, device="cuda")
    res = torch.randn(N, D, device="cuda").bfloat16()
    gamma, beta = torch.randn(D, device="cuda") * 0.5 + 1.0, torch.randn(D, device="cuda") * 0.1
    c = torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    stats = torch.empty(N, 2, device="cuda")
    ops.gemm(L.GEMM_NT, a, W, c, N, D, K, ops.epilogue(bias=b, res=res, alpha=0.5, ln_fwd=(gamma, beta, y, stats, 1e-5, act)))
    cref = res.double() + 0.5 * (a.double() @ W.double().t() + b.double())
    mean, var = cref.mean(1, keepdim=True), cref.var(1, unbiased=False, keepdim=True)
    yref = (cref - mean) * (var + 1e-5).rsqrt() * gamma.double() + beta.double()
    if act == 2:
        yref = yref * torch.sigmoid(yref)
    assert rel_err(c, cref) < 1e-2 and rel_err(y, yref) < 1e-2
    assert rel_err(stats[:, 0], mean[:, 0]) < 1e-3 and rel_err(stats[:, 1], (var + 1e-5).rsqrt()[:, 0]) < 1e-3


@pytest.mark.parametrize("D", [256, 512])
@pytest.mark.parametrize("N,K", [(4096 + 9, 1024), (300, 2048), (45000 + 3, 512)])
@pytest.mark.parametrize("layout", ["NT", "NN"])
def test_gemm_epilogue_layernorm_float32_stream(N, K, layout, D):
    """The autocast forms (float32 residual stream next to bf16 operands), both weight layouts: forward = Linear + bias +
    dropout-free alpha + float32 residual -> float32 stream tensor, LayerNorm appended (bf16 for the next GEMM, or float32 when
    the LayerNorm output is itself the stream: norm2); backward = dgrad + LayerNorm backward with a float32 LayerNorm input."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(N + K + D)
    a = torch.randn(N, K, device="cuda").bfloat16()
    W = (torch.randn(D, K, device="cuda") * 0.05).bfloat16()                 # (D, K): NT weight; NN takes its transpose (K, D)
    Wop, lay = (W, L.GEMM_NT) if layout == "NT" else (W.t().contiguous(), L.GEMM_NN)
    b = torch.randn(D, device="cuda")
    res = torch.randn(N, D, device="cuda")
    gamma, beta = torch.randn(D, device="cuda") * 0.5 + 1.0, torch.randn(D, device="cuda") * 0.1
    cref = res.double() + 0.5 * (a.double() @ W.double().t() + b.double())
    mean, var = cref.mean(1, keepdim=True), cref.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    yref = (cref - mean) * rstd * gamma.double() + beta.double()
    for ydt in (torch.bfloat16, torch.float32):
        c = torch.empty(N, D, device="cuda")
        y = torch.empty(N, D, device="cuda", dtype=ydt)
        stats = torch.empty(N, 2, device="cuda")
        ops.gemm(lay, a, Wop, c, N, D, K, ops.epilogue(bias=b, res=res, alpha=0.5, out_mode=L.OUT_F32, ln_fwd=(gamma, beta, y, stats, 1e-5, L.ACT_NONE)))
        assert rel_err(c, cref) < 1e-2 and rel_err(y, yref) < 1e-2, (ydt, rel_err(c, cref), rel_err(y, yref))
        assert rel_err(stats[:, 0], mean[:, 0]) < 2e-3 and rel_err(stats[:, 1], rstd[:, 0]) < 2e-3
    # backward: the LayerNorm input is the float32 stream, gradients bf16
    x32 = torch.randn(N, D, device="cuda") * 2 + 0.3
    xd = x32.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    st = torch.cat([mean, rstd], 1).float().contiguous()
    rg = torch.randn(N, D, device="cuda").bfloat16()
    tr = L.lib().smx_gemm_ln_tile_rows_for(N, D)      # (64-row tiles at d_model 256 up to ~38 000 frames: round 6)
    partial = torch.zeros((N + tr - 1) // tr, 2, D, device="cuda")
    dx = torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    ops.gemm(lay, a, Wop, dx, N, D, K, ops.epilogue(res=rg, ln_bwd=(x32, st, gamma, partial, None, None, None, True)))
    g = a.double() @ W.double().t()
    xh = (xd - mean) * rstd
    gg = g * gamma.double()
    ref = rstd * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)) + rg.double()
    assert rel_err(dx, ref) < 1e-2, rel_err(dx, ref)
    assert rel_err(partial[:, 0].sum(0), (g * xh).sum(0)) < 2e-3 and rel_err(partial[:, 1].sum(0), g.sum(0)) < 2e-3


@pytest.mark.parametrize("D", [256, 512])
@pytest.mark.parametrize("N,K", [(4096 + 9, 2048), (300, 512)])
def test_gemm_epilogue_layernorm_pair(N, K, D):
    """A Conformer layer's norm2 AND the next layer's first LayerNorm in the down-projection's epilogue (smx_epilogue.lnf2_*, the
    row-complete tiles on the float32 stream): C, y1 = LN1(C) (float32: the stream), y2 = LN2(y1) (bf16) and both statistics against
    float64; the shapes that cannot take it answer smx_gemm_ln_pair_ok = 0 and a call that is not the float32-stream form fails loudly."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(N + K)
    assert L.lib().smx_gemm_ln_pair_ok(L.BF16, N, D, K) == 1 and L.lib().smx_gemm_ln_pair_ok(L.BF16, N, 384, K) == 0
    a = torch.randn(N, K, device="cuda").bfloat16()
    W = (torch.randn(D, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(D, device="cuda")
    res = torch.randn(N, D, device="cuda")
    g1, b1 = torch.randn(D, device="cuda") * 0.5 + 1.0, torch.randn(D, device="cuda") * 0.1
    g2, b2 = torch.randn(D, device="cuda") * 0.5 + 1.0, torch.randn(D, device="cuda") * 0.1
    c, y1 = torch.empty(N, D, device="cuda"), torch.empty(N, D, device="cuda")
    y2 = torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    s1, s2 = torch.empty(N, 2, device="cuda"), torch.empty(N, 2, device="cuda")
    e = ops.epilogue(bias=b, res=res, alpha=0.5, drop=None, out_mode=L.OUT_F32, ln_fwd=(g1, b1, y1, s1, 1e-5, L.ACT_NONE),
                     ln_fwd2=(g2, b2, y2, s2, 1e-5))
    ops.gemm(L.GEMM_NT, a, W, c, N, D, K, e)
    cref = res.double() + 0.5 * (a.double() @ W.double().t() + b.double())
    m1, v1 = cref.mean(1, keepdim=True), cref.var(1, unbiased=False, keepdim=True)
    y1r = (cref - m1) * (v1 + 1e-5).rsqrt() * g1.double() + b1.double()
    m2, v2 = y1r.mean(1, keepdim=True), y1r.var(1, unbiased=False, keepdim=True)
    y2r = (y1r - m2) * (v2 + 1e-5).rsqrt() * g2.double() + b2.double()
    assert rel_err(c, cref) < 1e-2 and rel_err(y1, y1r) < 1e-2 and rel_err(y2, y2r) < 1.5e-2, (rel_err(c, cref), rel_err(y1, y1r), rel_err(y2, y2r))
    assert rel_err(s1[:, 0], m1[:, 0]) < 2e-3 and rel_err(s1[:, 1], (v1 + 1e-5).rsqrt()[:, 0]) < 2e-3
    assert rel_err(s2[:, 0], m2[:, 0]) < 2e-2 and rel_err(s2[:, 1], (v2 + 1e-5).rsqrt()[:, 0]) < 2e-3
    # not the float32-stream form (bf16 residual and output): refused, not silently skipped
    cb, rb = torch.empty(N, D, device="cuda", dtype=torch.bfloat16), res.bfloat16()
    with pytest.raises(RuntimeError):
        ops.gemm(L.GEMM_NT, a, W, cb, N, D, K, ops.epilogue(bias=b, res=rb, ln_fwd=(g1, b1, y2, s1, 1e-5, L.ACT_NONE), ln_fwd2=(g2, b2, y2, s2, 1e-5)))


@pytest.mark.parametrize("D", [256, 512])
def test_gemm_epilogue_layernorm_backward_with_activations(D):
    """The extended instantiation: the LayerNorm had a fused activation (y = swish(LN(x)), conv module LN2) and the second
    output carries the consumer's activation backward (dX2 = dX * swish'(z), the cell's merge)."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(11)
    N, K = 3000, 256
    dz = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(K, D, device="cuda") * 0.05).bfloat16()
    x = torch.randn(N, D, device="cuda").bfloat16()
    z2 = torch.randn(N, D, device="cuda").bfloat16()
    gamma, beta = torch.randn(D, device="cuda") * 0.5 + 1.0, torch.randn(D, device="cuda") * 0.3
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    stats = torch.cat([mean, rstd], 1).float().contiguous()
    tr = L.lib().smx_gemm_ln_tile_rows_for(N, D)      # (64-row tiles at d_model 256 up to ~38 000 frames: round 6)
    partial = torch.zeros((N + tr - 1) // tr, 2, D, device="cuda")
    dx, dx2 = torch.empty(N, D, device="cuda", dtype=torch.bfloat16), torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    e = ops.epilogue(ln_bwd=(x, stats, gamma, partial, dx2, (1.0, None, None, z2, L.ACT_SWISH), (beta, L.ACT_SWISH)))
    ops.gemm(L.GEMM_NN, dz, W, dx, N, D, K, e)

    def dswish(v):
        s = torch.sigmoid(v)
        return s * (1 + v * (1 - s))
    xh = (xd - mean) * rstd
    g = (dz.double() @ W.double()) * dswish(xh * gamma.double() + beta.double())
    gg = g * gamma.double()
    ref = rstd * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    assert rel_err(dx, ref) < 1e-2
    assert rel_err(dx2, ref * dswish(z2.double())) < 1.5e-2
    assert rel_err(partial[:, 0].sum(0), (g * xh).sum(0)) < 2e-3 and rel_err(partial[:, 1].sum(0), g.sum(0)) < 2e-3


@pytest.mark.parametrize("chunk,D,k", [(4, 256, 31), (8, 256, 31), (16, 256, 31), (0, 40, 7), (6, 72, 15), (0, 64, 31)])
def test_chunked_dwconv_backward_is_bit_reproducible(chunk, D, k):
    """Dynamic Chunk Convolution in the rolling kernels: per-wave partial tap gradients reduced in a fixed order - two runs
    on the same inputs give bit-identical dP, dW and dbias (no atomics), at D = 256, k = 31 (the Conformer shape); the
    generic tiled kernel (other k / D) goes through per-workgroup partial rows as well since round 3."""
    L, ops = _ops()
    torch.manual_seed(chunk + k)
    B, T = 8, 500
    p = torch.randn(B * T, 2 * D, device="cuda").bfloat16()
    dy = torch.randn(B * T, D, device="cuda").bfloat16()
    w = torch.randn(D, k, device="cuda") * 0.3
    bias = torch.randn(D, device="cuda")
    outs = []
    for _ in range(2):
        dw, db = torch.zeros(D, k, device="cuda"), torch.zeros(D, device="cuda")
        dp, _ = ops.dwconv_bwd(dy, p, w, bias, dw, db, B, T, D, k, True, L.PAD_ZERO, chunk)
        torch.cuda.synchronize()
        outs.append((dp.clone(), dw.clone(), db.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    y1 = ops.dwconv_fwd(p, w, bias, B, T, D, k, True, L.PAD_ZERO, chunk).clone()
    y2 = ops.dwconv_fwd(p, w, bias, B, T, D, k, True, L.PAD_ZERO, chunk)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("N,D", [(1, 64), (37, 144), (1000, 256), (3750, 512), (513, 1024), (70, 2048)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_layernorm_pair_equals_two_launches(N, D, out_dtype):
    """smx_layernorm_fwd_pair_x32 (norm2 of a Conformer layer + the next layer's first LayerNorm, Conformer.py:536 + :458-459, in one
    pass over the float32 stream) against smx_layernorm_fwd followed by smx_layernorm_fwd_x32 (outputs and statistics to an ulp)
    and against torch's LayerNorm."""
    from summarymixing_amd import ops
    torch.manual_seed(N + D)
    x = (torch.randn(N, D, device="cuda") * 3 + 0.7)
    g1, b1 = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
    g2, b2 = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
    assert ops.layernorm_pair_ok(x, out_dtype)
    y1, s1, y2, s2 = ops.layernorm_fwd_pair(x, g1, b1, 1e-5, g2, b2, 1e-6, True, out_dtype)
    r1, t1 = ops.layernorm_fwd(x, g1, b1, 1e-5, True)
    r2, t2 = ops.layernorm_fwd(r1, g2, b2, 1e-6, True, out_dtype=out_dtype)
    # (the two code paths contract their multiply-adds differently under -ffast-math: equal to an ulp, not bit for bit)
    assert rel_err(y1, r1) < 1e-6 and rel_err(s1, t1) < 1e-6 and rel_err(s2, t2) < 1e-6
    assert rel_err(y2, r2) < (1e-6 if out_dtype == torch.float32 else 2.0 ** -7)
    ref1 = torch.nn.functional.layer_norm(x.double(), (D,), g1.double(), b1.double(), 1e-5)
    ref2 = torch.nn.functional.layer_norm(ref1, (D,), g2.double(), b2.double(), 1e-6)
    assert rel_err(y1, ref1) < 1e-5
    assert rel_err(y2, ref2) < (1e-5 if out_dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("mode", ["2", "3"])
def test_gemm_256x256_tile_every_eligible_shape(mode):
    """SMX_T256=2 (read once per process, hence a child): EVERY eligible shape takes the 256 x 256 tile - the residual, saved
    pre-activation, float32-output, activation-gradient and column-sum epilogues on a 256-row tile with a ragged tail, NT and
    NN, K = 512 and K = 2048, against float64 references.  SMX_T256=3: the M = 512 shapes of the same list on the row-complete
    128 x 512 tile of the LayerNorm-fused kernels (its plain-epilogue instantiation)."""
    import os
    import subprocess
    import sys
    script = r'''
import os, sys, torch
sys.path.insert(0, os.environ["SMX_ROOT"])
from summarymixing_amd import _lib as L, ops
assert L.get_config()["t256"] == int(os.environ["SMX_T256"])
torch.manual_seed(0)
rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
for N, K, M in ((25600 + 77, 2048, 512), (1000, 512, 256), (4096 + 5, 512, 768), (300, 128, 512), (5000 + 3, 1024, 512)):
    x = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.03).bfloat16()
    b = torch.randn(M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.2).to(torch.uint8)
    res = torch.randn(N, M, device="cuda").bfloat16()
    res32 = torch.randn(N, M, device="cuda")
    zr = x.double() @ w.double().t() + b.double()
    y, z = torch.empty(N, M, device="cuda", dtype=torch.bfloat16), torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z, row_mask=mask, res=res, alpha=0.5))
    assert rel(z, zr) < 1e-2, (N, K, M, "Z")
    assert rel(y, res.double() + 0.5 * torch.nn.functional.silu(zr) * mask[:, None]) < 1e-2, (N, K, M, "NT res + swish + Z")
    y32 = torch.empty(N, M, device="cuda", dtype=torch.float32)
    ops.gemm(L.GEMM_NT, x, w, y32, N, M, K, ops.epilogue(bias=b, res=res32, alpha=0.5, out_mode=L.OUT_F32))
    assert rel(y32, res32.double() + 0.5 * zr) < 1e-2, (N, K, M, "NT fp32 residual stream")
    wt = w.t().contiguous()
    ops.gemm(L.GEMM_NN, x, wt, y, N, M, K, ops.epilogue(res=res))
    assert rel(y, x.double() @ wt.double() + res.double()) < 1e-2, (N, K, M, "NN res")
    zz = torch.randn(N, M, device="cuda").bfloat16()
    cs = torch.zeros(M, device="cuda")
    ops.gemm(L.GEMM_NN, x, wt, y, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=zz, colsum=cs))
    zf = zz.double(); sg = torch.sigmoid(zf)
    assert rel(y, (x.double() @ wt.double()) * (sg * (1 + zf * (1 - sg)))) < 1e-2, (N, K, M, "act-grad")
    ysum = y.double().sum(0)
    assert float((cs.double() - ysum).abs().max() / ysum.abs().max()) < 2e-2, (N, K, M, "colsum")
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, SMX_T256=mode, SMX_ROOT=root), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,D", [(10, 375, 512), (1, 500, 256), (3, 17, 40), (7, 1300, 192), (2, 4096, 64), (128, 500, 256)])
def test_pool_bcast_equals_masked_mean_then_broadcast(dtype, B, T, D):
    """smx_pool_bcast (round 6, small batches: one launch) against the three-launch path it replaces: the masked mean, its inverse
    count, the dropped broadcast (bit-identical keep decisions: the mask is a function of the element index) and the act / mask
    backward form - forward also IN PLACE on a column slice (the fast mode's cat buffer, summary_mixing.py:257-267,282-284)."""
    L, ops = _ops()
    assert ops.pool_bcast_ok(B, T, D)
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    big = (torch.rand(B * T, D + 24, device="cuda", generator=g) * 2 - 1).to(dtype)
    s = big[:, 24:] if D % 8 == 0 else big[:, 3:3 + D]
    mask = (torch.rand(B * T, device="cuda", generator=g) > 0.3).to(torch.uint8)
    mask.view(B, T)[:, 0] = 1
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    # the mean and the inverse count
    m_ref, inv_ref = ops.masked_mean(s, mask, B, T, scale=True, want_inv=True)
    m1, inv1 = ops.pool_bcast(s, mask, B, T, ds=None, scale=True, want_mean=True, want_inv=True)
    ref64 = (s.double().view(B, T, D) * mask.view(B, T, 1)).sum(1) / mask.view(B, T).sum(1, keepdim=True).double()
    assert rel_err(m1, ref64) < 2e-6 and rel_err(m1, m_ref) < 2e-6 and torch.equal(inv1, inv_ref)
    # forward: repeat + dropout, out of place and in place
    out_ref = torch.empty(B * T, D, device="cuda", dtype=dtype)
    ops.bcast_rows(m_ref, None, out_ref, B, T, drop=(0.2, 4242))
    out = torch.full((B * T, D), 9.0, device="cuda", dtype=dtype)
    ops.pool_bcast(s, mask, B, T, ds=out, scale=True, want_mean=False, drop=(0.2, 4242))
    assert torch.equal(out == 0, out_ref == 0) and rel_err(out, out_ref) < tol
    if D % 8 == 0:
        keep = s.clone()
        ops.pool_bcast(s, mask, B, T, ds=s, scale=True, want_mean=False, drop=(0.2, 4242))
        assert torch.equal(s, out), "in place differs from out of place"
        s.copy_(keep)
    # backward: sum over time, * inv, * act'(z) * mask
    z = (torch.rand(B * T, D, device="cuda", generator=g) * 6 - 3).to(dtype)
    dsum, _ = ops.masked_mean(s, None, B, T, scale=False)
    d_ref = torch.empty(B * T, D, device="cuda", dtype=dtype)
    ops.bcast_rows_act_bwd(dsum, inv_ref, d_ref, B, T, z, mask, L.ACT_SWISH)
    d1 = torch.full((B * T, D), 9.0, device="cuda", dtype=dtype)
    ops.pool_bcast(s, None, B, T, ds=d1, scale=False, want_mean=False, inv_in=inv_ref, z=z, mask_out=mask, act=L.ACT_SWISH)
    assert rel_err(d1, d_ref) < tol and bool((d1.view(B, T, D)[mask.view(B, T) == 0] == 0).all())
    d2 = torch.empty_like(d1)
    ops.bcast_rows(dsum, inv_ref, d_ref, B, T)
    ops.pool_bcast(s, None, B, T, ds=d2, scale=False, want_mean=False, inv_in=inv_ref)
    assert rel_err(d2, d_ref) < tol
    # deterministic
    d3 = torch.empty_like(d1)
    ops.pool_bcast(s, None, B, T, ds=d3, scale=False, want_mean=False, inv_in=inv_ref)
    assert torch.equal(d2, d3)
    assert ops.pool_bcast_ok(128, 500, 256) and not ops.pool_bcast_ok(8, 30000, 512) and not ops.pool_bcast_ok(20, 2000, 256)


def test_reduce_jobs_many_jobs_and_ragged_tails():
    """smx_reduce_jobs after the round-6 rewrite (4 groups per thread, ballot job lookup): more than 64 jobs, vector and scalar
    jobs, source counts on every lane-sharing branch (1, 2, 7, 20), sizes that do not fill the last workgroup."""
    import ctypes
    L, ops = _ops()
    torch.manual_seed(11)
    specs = [(2, 256, 512), (1, 1, 512), (7, 300, 64), (20, 33, 128), (2, 5, 7), (3, 1, 5), (16, 1, 1024)] + [(2, 64, 64)] * 70
    arr = (L.ReduceJob * len(specs))()
    srcs, dsts, refs, starts = [], [], [], [0]
    for i, (ns, rows, cols) in enumerate(specs):
        src = torch.randn(ns, rows, cols, device="cuda")
        dst = torch.randn(rows, cols, device="cuda")
        refs.append(dst.double() + 0.5 * src.double().sum(0))
        vec = int(cols % 4 == 0)
        arr[i] = L.ReduceJob(src.data_ptr(), dst.data_ptr(), rows * cols, cols, ns, rows, cols, 0.5, vec, 0)
        starts.append(starts[-1] + L.lib().smx_reduce_job_blocks(ctypes.byref(arr[i])))
        srcs.append(src); dsts.append(dst)
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    starts_dev = torch.tensor(starts, dtype=torch.int32).cuda()
    ops.reduce_jobs(jobs_dev, starts_dev, len(specs), starts[-1])
    for d, r in zip(dsts, refs):
        assert rel_err(d, r) < 1e-6


@pytest.mark.parametrize("rows", [3750, 500, 64, 1, 129, 8192 + 17])
def test_wgrad_group_direct_matches_float64(rows):
    """smx_wgrad_group_direct (round 6: the weight gradients of a layer without split-K slabs, added INTO the gradients): against
    float64, accumulation on top of what the gradient buffers hold, the bias gradients, a ragged frame count, strided views, and
    bit-identical repeats."""
    L, ops = _ops()
    torch.manual_seed(rows)
    shapes = [(512, 256), (256, 512), (128, 128), (256, 256), (1024, 128)]
    big_z = (torch.randn(rows, 2304, device="cuda") * 0.5).bfloat16()
    recs, refs, off = [], [], 0
    for i, (M, K) in enumerate(shapes):
        dz = big_z[:, off:off + M]; off += M                      # column slices of one buffer: leading dimension > width
        x = torch.randn(rows, K, device="cuda").bfloat16()
        gbuf = torch.randn(M, K + 64, device="cuda")
        gW = gbuf[:, 32:32 + K]                                    # (a strided float32 gradient view, 16-byte aligned)
        gb = torch.randn(M, device="cuda") if i != 2 else None
        refs.append((gW.double() + dz.double().t() @ x.double(), (gb.double() + dz.double().sum(0)) if gb is not None else None, gbuf.clone()))
        recs.append((dz, x, gW, gb, M, K))
    ops.wgrad_group_direct(recs, rows)
    for (dz, x, gW, gb, M, K), (rw, rb, g0) in zip(recs, refs):
        assert rel_err(gW, rw) < 2e-6, (M, K, rel_err(gW, rw))
        if gb is not None:
            assert rel_err(gb, rb) < 2e-6
        big = gW._base if gW._base is not None else gW
        assert torch.equal(big[:, :32], g0[:, :32]) and torch.equal(big[:, 32 + K:], g0[:, 32 + K:])    # nothing outside the view is touched
    # bit-identical when repeated from the same starting point
    outs = []
    for _ in range(2):
        fresh = [(dz, x, torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda"), M, K) for dz, x, _, _, M, K in recs]
        ops.wgrad_group_direct(fresh, rows)
        outs.append([f[2] for f in fresh] + [f[3] for f in fresh])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
