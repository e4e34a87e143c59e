"""The panel-resident GEMM (smx_gemm_panel + smx_weight_pack, csrc/gemm_panel.h) against float64 torch math and against the
tiled smx_gemm doing the same work: FFN up-projection forward (bias, activation, saved pre-activation, dropout) and the act-grad
dgrad of the down-projection (Conformer.py:458-472 and its autograd backward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from summarymixing_amd import _lib as L, ops      # noqa: E402
from tests._util import rel_err                   # noqa: E402

ACTS = {L.ACT_NONE: (lambda v: v), L.ACT_SWISH: torch.nn.functional.silu, L.ACT_GELU: torch.nn.functional.gelu,
        L.ACT_RELU: torch.relu}


def _mk(N, M, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1
    return rnd(N, K).bfloat16(), (rnd(M, K) * (2.0 / K ** 0.5)).bfloat16(), rnd(M) * 0.3


# (N < 24 576 with M >= 1024: the chunk rounds of a panel are dealt to 2 or 4 workgroups; 16 000 x 1024 and 8 000 x 2048 fill the chip that way)
@pytest.mark.parametrize("N,M,K", [(1000 + 37, 1024, 256), (128, 512, 256), (1, 64, 256), (700, 2048, 512), (391, 1536, 512), (2500, 3072, 512),
                                   (16000 + 5, 1024, 256), (8000 + 9, 2048, 512),
                                   (3750, 2048, 512), (6000 + 1, 1024, 256), (3750, 1024, 512)])   # (round 6: 64- and 32-row panels, the recipe batch)
@pytest.mark.parametrize("act", [L.ACT_SWISH, L.ACT_GELU, L.ACT_RELU, L.ACT_NONE])
def test_panel_forward_matches_float64(N, M, K, act):
    x, W, b = _mk(N, M, K)
    assert ops.gemm_panel_ok(x, M, K)
    wp = ops.weight_pack(W, bias=b)
    out, z = torch.full((N, M), 7.0, device="cuda").bfloat16(), torch.full((N, M), 7.0, device="cuda").bfloat16()
    ops.gemm_panel(x, wp, out, N, M, K, ops.epilogue(act=act, z=z))
    zr = x.double() @ W.double().t() + b.double()
    assert rel_err(z, zr) < 1e-2
    # the activation runs on the bf16-rounded pre-activation (autocast semantics): exact against the kernel's own Z up to rounding
    assert rel_err(out, ACTS[act](z.double())) < 5e-3
    assert rel_err(out, ACTS[act](zr)) < 1e-2
    # ... and against the tiled kernel (activation on the float32 accumulator there)
    o2, z2 = torch.empty_like(out), torch.empty_like(z)
    ops.gemm(L.GEMM_NT, x, W, o2, N, M, K, ops.epilogue(bias=b, act=act, z=z2))
    e1, e2 = rel_err(z, z2), rel_err(out, o2)
    assert e1 < 8e-3 and e2 < 1.2e-2, (e1, e2)           # (one bf16 ulp of the largest element: the two kernels round different sums)
    # no bias / no saved Z
    o3 = torch.empty_like(out)
    ops.gemm_panel(x, ops.weight_pack(W), o3, N, M, K, ops.epilogue(act=act))
    assert rel_err(o3, ACTS[act](x.double() @ W.double().t())) < 1e-2


@pytest.mark.parametrize("N,M,K", [(1000 + 37, 1024, 256), (5, 128, 256), (700, 2048, 512), (2500, 3072, 512), (16000 + 5, 1024, 256),
                                   (3750, 2048, 512), (6000 + 1, 1024, 256)])
@pytest.mark.parametrize("act", [L.ACT_SWISH, L.ACT_GELU, L.ACT_RELU])
def test_panel_actgrad_matches_float64(N, M, K, act):
    dy, Wt, _ = _mk(N, M, K, seed=1)           # Wt (M, K) = W2^T
    W2 = Wt.t().contiguous()                   # (K, M): the down-projection's weight as its dgrad sees it
    z = (torch.rand(N, M, device="cuda") * 6 - 3).bfloat16()
    wp = ops.weight_pack(W2, transposed=True)
    assert torch.equal(wp, ops.weight_pack(Wt))                      # the packing kernel's transposition
    dz = torch.full((N, M), 7.0, device="cuda").bfloat16()
    ops.gemm_panel(dy, wp, dz, N, M, K, ops.epilogue(act=act, act_grad_z=z))
    zd = z.double().requires_grad_(True)
    (ACTS[act](zd)).sum().backward()
    ref = (dy.double() @ W2.double()) * zd.grad
    assert rel_err(dz, ref) < 1.2e-2
    d2 = torch.empty_like(dz)
    ops.gemm(L.GEMM_NN, dy, W2, d2, N, M, K, ops.epilogue(act=act, act_grad_z=z))
    assert rel_err(dz, d2) < 1.2e-2


@pytest.mark.parametrize("K,M", [(256, 1024), (512, 2048)])
def test_panel_dropout_mask_is_the_tiled_kernels(K, M):
    """Same seed -> the same keep decisions as smx_gemm (the backward of a layer may run on either kernel)."""
    N = 777
    x, W, b = _mk(N, M, K, seed=2)
    wp = ops.weight_pack(W, bias=b)
    o1, o2 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16), torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    ops.gemm_panel(x, wp, o1, N, M, K, ops.epilogue(act=L.ACT_NONE, drop=(0.15, 1234)))
    ops.gemm(L.GEMM_NT, x, W, o2, N, M, K, ops.epilogue(bias=b, act=L.ACT_NONE, drop=(0.15, 1234)))
    assert torch.equal(o1 == 0, o2 == 0)
    assert 0.13 < float((o1 == 0).float().mean()) < 0.17
    assert rel_err(o1, o2) < 8e-3
    # act-grad form: the same mask again
    z = (torch.rand(N, M, device="cuda") * 6 - 3).bfloat16()
    dy = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    d1, d2 = torch.empty_like(o1), torch.empty_like(o1)
    W2 = W.t().contiguous()
    ops.gemm_panel(dy, ops.weight_pack(W2, transposed=True), d1, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 1234)))
    ops.gemm(L.GEMM_NN, dy, W2, d2, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 1234)))
    assert torch.equal((d1 == 0) | (d2 == 0), o1 == 0) or float(((d1 == 0) != (o1 == 0)).float().mean()) < 1e-3   # (exact zeros of the product aside)
    assert rel_err(d1, d2) < 1.2e-2


def test_panel_strided_views_and_padding_rows_untouched():
    """Column slices as operands (leading dimension > width); rows beyond N of the output buffer are not written."""
    N, M, K = 300, 512, 256
    big_x = (torch.rand(N, K + 64, device="cuda") * 2 - 1).bfloat16()
    x = big_x[:, 64:]
    _, W, b = _mk(N, M, K, seed=3)
    big_o = torch.full((N + 50, M + 128), 3.0, device="cuda").bfloat16()
    out = big_o[:N, 128:]
    ops.gemm_panel(x, ops.weight_pack(W, bias=b), out, N, M, K, ops.epilogue(act=L.ACT_SWISH))
    ref = torch.nn.functional.silu(x.double() @ W.double().t() + b.double())
    assert rel_err(out, ref) < 1e-2
    assert bool((big_o[N:] == 3.0).all()) and bool((big_o[:, :128] == 3.0).all())


def test_panel_refuses_what_it_cannot_do():
    N, M, K = 256, 512, 256
    x, W, b = _mk(N, M, K)
    out = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    wp = ops.weight_pack(W)
    with pytest.raises(RuntimeError):
        ops.gemm_panel(x, wp, out, N, M, K, ops.epilogue(res=out))
    with pytest.raises(RuntimeError):
        ops.gemm_panel(x, wp, out, N, M, K, ops.epilogue(bias=b))
    with pytest.raises(RuntimeError):
        ops.gemm_panel(x, wp, out, N, M, K, ops.epilogue(out_mode=L.OUT_F32))
    assert not ops.gemm_panel_ok(x, M, 384) and not ops.gemm_panel_ok(x, 96, K) and not ops.gemm_panel_ok(x.float(), M, K)


@pytest.mark.parametrize("K,M", [(256, 1024), (512, 2048), (512, 1536)])
def test_panel_is_deterministic_at_full_chip_sizes(K, M):
    """More workgroups than CUs, several chunk rounds per wave: every run bit-identical to the first, and equal to the tiled kernel
    within a bf16 ulp (a hand-counted-vmcnt version of this kernel passed every small test and produced garbage here)."""
    N = 40000 + 77
    x, W, b = _mk(N, M, K, seed=4)
    z = (torch.rand(N, M, device="cuda") * 6 - 3).bfloat16()
    Wt = W.t().contiguous()
    wp, wpt = ops.weight_pack(W, bias=b), ops.weight_pack(Wt, transposed=True)
    cases = [(lambda o, zz: ops.gemm_panel(x, wp, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, z=zz, drop=(0.15, 5))),
              lambda o, zz: ops.gemm(L.GEMM_NT, x, W, o, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=zz, drop=(0.15, 5)))),
             (lambda o, zz: ops.gemm_panel(x, wpt, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 5))),
              lambda o, zz: ops.gemm(L.GEMM_NN, x, Wt, o, N, M, K, ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 5)))),
             (lambda o, zz: ops.gemm_panel(x, wpt, o, N, M, K, ops.epilogue()), lambda o, zz: ops.gemm(L.GEMM_NN, x, Wt, o, N, M, K, ops.epilogue()))]
    for fp, ft in cases:
        o0, z0, ot, zt = (torch.empty(N, M, device="cuda", dtype=torch.bfloat16) for _ in range(4))
        fp(o0, z0)
        ft(ot, zt)
        assert rel_err(o0, ot) < 1.2e-2
        for _ in range(4):
            o1, z1 = torch.full_like(o0, 3.0), torch.full_like(o0, 3.0)
            fp(o1, z1)
            assert torch.equal(o1, o0)


def test_weight_pack_jobs_equals_single_packs():
    """The grouped re-pack (one launch for a table of weights) writes what the single calls write."""
    import ctypes
    torch.manual_seed(5)
    specs = [(1024, 256, False, True), (256, 1024, True, False), (512, 256, False, True), (2048, 512, False, False), (512, 1536, True, False)]
    ws, outs, refs, biases = [], [], [], []
    arr, nb = (L.PackJob * len(specs))(), 0
    for j, (r, c, tr, hb) in zip(arr, specs):
        W = (torch.rand(r, c, device="cuda") - 0.5).bfloat16()
        b = torch.rand(c if tr else r, device="cuda") if hb else None
        M, K = (c, r) if tr else (r, c)
        ref = ops.weight_pack(W, transposed=tr, bias=b)
        out = torch.zeros_like(ref)
        j.W, j.ldw, j.bias, j.packed, j.M, j.K, j.transposed, j.block_start = W.data_ptr(), W.stride(0), (b.data_ptr() if hb else None), out.data_ptr(), M, K, int(tr), nb
        nb += L.lib().smx_weight_pack_job_blocks(M, K)
        ws.append(W); outs.append(out); refs.append(ref); biases.append(b)
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    ops.weight_pack_jobs(dev, len(specs), nb)
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


@pytest.mark.parametrize("N,M,K", [(1000 + 37, 512, 256), (3000, 1024, 256), (700, 512, 512), (20000 + 3, 1536, 512)])
def test_panel_row_mask_and_alpha(N, M, K):
    """C = alpha * D(act(.)) * row_mask in both forms (the VanillaNN layers of the cell, summary_mixing.py:207-210): masked rows
    are exact zeros (they enter the panel as zeros and get no bias), the others equal the tiled kernel's."""
    x, W, b = _mk(N, M, K, seed=6)
    mask = (torch.rand(N, device="cuda") > 0.3).to(torch.uint8)
    mask[-5:] = 0
    z = (torch.rand(N, M, device="cuda") * 6 - 3).bfloat16()
    Wt = W.t().contiguous()
    o1, o2, z1, z2 = (torch.full((N, M), 5.0, device="cuda").bfloat16() for _ in range(4))
    ops.gemm_panel(x, ops.weight_pack(W, bias=b), o1, N, M, K, ops.epilogue(act=L.ACT_SWISH, z=z1, row_mask=mask, alpha=0.5, drop=(0.15, 9)))
    ops.gemm(L.GEMM_NT, x, W, o2, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z2, row_mask=mask, alpha=0.5, drop=(0.15, 9)))
    assert bool((o1[mask == 0] == 0).all()) and rel_err(o1, o2) < 1.2e-2
    assert rel_err(z1[mask != 0], z2[mask != 0]) < 8e-3          # (the saved pre-activation of a masked row is never used: its gradient is masked too)
    ops.gemm_panel(x, ops.weight_pack(Wt, transposed=True), o1, N, M, K, ops.epilogue(act=L.ACT_GELU, act_grad_z=z, row_mask=mask, alpha=0.5, drop=(0.15, 9)))
    ops.gemm(L.GEMM_NN, x, Wt, o2, N, M, K, ops.epilogue(act=L.ACT_GELU, act_grad_z=z, row_mask=mask, alpha=0.5, drop=(0.15, 9)))
    assert bool((o1[mask == 0] == 0).all()) and rel_err(o1, o2) < 1.2e-2


def test_panel_dropout_on_leading_columns_only():
    """drop_cols: dropout on the first drop_cols output columns, mask index n * drop_cols + m (the cell's local | summary projection)."""
    N, M, K, dc = 2000, 512, 256, 256
    x, W, b = _mk(N, M, K, seed=7)
    o1, o2 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16), torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    ops.gemm_panel(x, ops.weight_pack(W, bias=b), o1, N, M, K, ops.epilogue(act=L.ACT_SWISH, drop=(0.25, 77), drop_cols=dc))
    ops.gemm(L.GEMM_NT, x, W, o2, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, drop=(0.25, 77), drop_cols=dc))
    assert torch.equal(o1[:, :dc] == 0, o2[:, :dc] == 0) and 0.2 < float((o1[:, :dc] == 0).float().mean()) < 0.3
    assert float((o1[:, dc:] == 0).float().mean()) < 0.01 and rel_err(o1, o2) < 1.2e-2


@pytest.mark.parametrize("rows", [128, 64, 32])
def test_panel_every_panel_height_in_a_subprocess(rows):
    """SMX_PANEL_ROWS forces the panel height (read once per process): both forms, with row mask, dropout and a ragged last panel, against
    the tiled kernel - bit-identical dropout masks, values within a bf16 ulp - and run twice for determinism."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from summarymixing_amd import _lib as L, ops
from tests._util import rel_err
assert L.get_config()["panel_rows"] == %d
g = torch.Generator(device="cuda").manual_seed(3)
rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1
for N, M, K in ((1000 + 37, 1024, 256), (3750, 2048, 512), (333, 512, 512), (70, 192, 256)):
    x, W, b = rnd(N, K).bfloat16(), (rnd(M, K) * (2.0 / K ** 0.5)).bfloat16(), rnd(M) * 0.3
    mask = (torch.rand(N, device="cuda", generator=g) > 0.3).to(torch.uint8)
    z = (rnd(N, M) * 3).bfloat16()
    Wt = W.t().contiguous()
    o1, o2, o3, z1, z2 = (torch.full((N, M), 5.0, device="cuda").bfloat16() for _ in range(5))
    ep = lambda **kw: ops.epilogue(row_mask=mask, alpha=0.5, drop=(0.15, 9), **kw)
    ops.gemm_panel(x, ops.weight_pack(W, bias=b), o1, N, M, K, ep(act=L.ACT_SWISH, z=z1))
    ops.gemm_panel(x, ops.weight_pack(W, bias=b), o3, N, M, K, ep(act=L.ACT_SWISH, z=z2))
    ops.gemm(L.GEMM_NT, x, W, o2, N, M, K, ep(bias=b, act=L.ACT_SWISH, z=z2))
    assert torch.equal(o1, o3) and torch.equal(o1 == 0, o2 == 0) and rel_err(o1, o2) < 1.2e-2, (N, M, K, rel_err(o1, o2))
    assert rel_err(z1[mask != 0], z2[mask != 0]) < 8e-3
    ops.gemm_panel(x, ops.weight_pack(Wt, transposed=True), o1, N, M, K, ep(act=L.ACT_GELU, act_grad_z=z))
    ops.gemm(L.GEMM_NN, x, Wt, o2, N, M, K, ep(act=L.ACT_GELU, act_grad_z=z))
    assert bool((o1[mask == 0] == 0).all()) and rel_err(o1, o2) < 1.2e-2, (N, M, K, rel_err(o1, o2))
print("ok")
""" % (root, rows)
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, SMX_PANEL_ROWS=str(rows)), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
