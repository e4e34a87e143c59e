"""Split-K over workgroups for the long reductions of a small batch (round 6): smx_gemm_panel_slabs + smx_slab_epilogue against the
tiled smx_gemm running the SAME smx_epilogue (bias, activation + saved Z, dropout, alpha, row mask, float32 residual stream, the
LayerNorm(s) appended: Conformer.py:458-476,507,530-536) and against float64 torch math; smx_layernorm_bwd with the gradient taken
from float32 slabs (the dgrad of the Linear behind a LayerNorm) against the two-launch path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from summarymixing_amd import _lib as L, ops      # noqa: E402
from tests._util import rel_err                   # noqa: E402


def _mk(N, M, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1
    return rnd(N, K).bfloat16(), (rnd(M, K) * (2.0 / K ** 0.5)).bfloat16(), rnd(M) * 0.3, g


@pytest.mark.parametrize("N,M,K,ks", [(3750, 512, 2048, 512), (3750, 512, 1024, 512), (500, 256, 1024, 256), (1000 + 37, 256, 512, 256),
                                      (3750, 512, 512, 512), (70, 192, 512, 256), (6000 + 1, 512, 2048, 512), (15000, 512, 2048, 512)])
def test_slabs_sum_to_the_product(N, M, K, ks):
    x, W, _, _ = _mk(N, M, K)
    ns = K // ks
    assert ops.panel_slabs_ok(x, M, ks, ns)
    slabs = torch.full((ns, N, M), 7.0, device="cuda")
    ops.gemm_panel_slabs(x, ops.weight_pack_slices(W, ks), slabs, N, M, ks, ns)
    for s_ in range(ns):
        ref = x[:, s_ * ks:(s_ + 1) * ks].double() @ W[:, s_ * ks:(s_ + 1) * ks].double().t()
        assert rel_err(slabs[s_], ref) < 2e-6, (s_, rel_err(slabs[s_], ref))
    # the dgrad orientation: W (K, M) packed transposed, slices = row ranges
    Wt = W.t().contiguous()
    s2 = torch.empty_like(slabs)
    ops.gemm_panel_slabs(x, ops.weight_pack_slices(Wt, ks, transposed=True), s2, N, M, ks, ns)
    assert torch.equal(s2, slabs)
    # deterministic
    s3 = torch.empty_like(slabs)
    ops.gemm_panel_slabs(x, ops.weight_pack_slices(W, ks), s3, N, M, ks, ns)
    assert torch.equal(s3, slabs)


@pytest.mark.parametrize("N,M,K,ks", [(3750, 512, 2048, 512), (500, 256, 1024, 256), (1000 + 37, 256, 512, 256), (333, 512, 512, 512)])
@pytest.mark.parametrize("variant", ["down", "merge", "convout", "pair", "plain_bf16"])
def test_slab_epilogue_equals_the_gemm_epilogue(N, M, K, ks, variant):
    """The same smx_epilogue through smx_gemm (tiled kernel, LayerNorm as a separate launch) and through slabs + smx_slab_epilogue."""
    x, W, b, g = _mk(N, M, K, seed=3)
    ns = K // ks
    res = torch.randn(N, M, device="cuda", generator=g)
    mask = (torch.rand(N, device="cuda", generator=g) > 0.3).to(torch.uint8)
    g1, b1 = torch.rand(M, device="cuda", generator=g) + 0.5, torch.randn(M, device="cuda", generator=g) * 0.1
    g2, b2 = torch.rand(M, device="cuda", generator=g) + 0.5, torch.randn(M, device="cuda", generator=g) * 0.1
    slabs = torch.empty((ns, N, M), device="cuda")
    ops.gemm_panel_slabs(x, ops.weight_pack_slices(W, ks), slabs, N, M, ks, ns)

    def run(kind):
        f32 = variant != "plain_bf16"
        c = torch.full((N, M), 5.0, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        z = torch.full((N, M), 5.0, device="cuda").bfloat16() if variant == "merge" else None
        y = torch.full((N, M), 5.0, device="cuda", dtype=torch.float32 if variant == "pair" else torch.bfloat16)
        y2 = torch.full((N, M), 5.0, device="cuda").bfloat16()
        st, st2 = torch.zeros(N, 2, device="cuda"), torch.zeros(N, 2, device="cuda")
        kw = dict(bias=b, out_mode=L.OUT_F32 if f32 else L.OUT_T)
        if variant == "down":
            kw.update(res=res, alpha=0.5, drop=(0.15, 99))
        elif variant == "merge":
            kw.update(res=res, act=L.ACT_SWISH, z=z)
        elif variant == "convout":
            kw.update(res=res, row_mask=mask, drop=(0.15, 77))
        elif variant == "pair":
            kw.update(res=res, alpha=0.5, drop=(0.15, 99))
        else:
            kw.update(res=res.bfloat16(), alpha=0.5)
        if kind == "slab":
            lnf = (g1, b1, y, st, 1e-5, L.ACT_NONE) if variant != "plain_bf16" else None
            lnf2 = (g2, b2, y2, st2, 1e-5) if variant == "pair" else None
            ops.slab_epilogue(slabs, ns, c, N, M, ops.epilogue(ln_fwd=lnf, ln_fwd2=lnf2, **kw))
        else:
            ops.gemm(L.GEMM_NT, x, W, c, N, M, K, ops.epilogue(**kw))
            if variant != "plain_bf16":
                if variant == "pair":
                    y1_, s1_, y2_, s2_ = ops.layernorm_fwd_pair(c, g1, b1, 1e-5, g2, b2, 1e-5, True, torch.bfloat16)
                    y.copy_(y1_); st.copy_(s1_); y2.copy_(y2_); st2.copy_(s2_)
                else:
                    yy, ss = ops.layernorm_fwd(c, g1, b1, 1e-5, True, L.ACT_NONE, out_dtype=torch.bfloat16)
                    y.copy_(yy); st.copy_(ss)
        return c, z, y, y2, st, st2

    cs, zs, ys, y2s, sts, st2s = run("slab")
    ct, zt, yt, y2t, stt, st2t = run("gemm")
    tol = 2e-5 if variant != "plain_bf16" else 8e-3
    if variant in ("down", "convout", "pair"):
        dropped = lambda c_: (c_ - res).abs() < 1e-12
        assert float((dropped(cs) != dropped(ct)).float().mean()) < 1e-4          # the same keep decisions (exact-zero products aside)
    assert rel_err(cs, ct) < tol, rel_err(cs, ct)
    if zs is not None:
        assert rel_err(zs, zt) < 8e-3
    if variant != "plain_bf16":
        assert rel_err(ys, yt) < (1e-4 if variant == "pair" else 1e-2), rel_err(ys, yt)
        assert rel_err(sts, stt) < 1e-4
        ref = torch.nn.functional.layer_norm(cs.double(), (M,), g1.double(), b1.double(), 1e-5)
        assert rel_err(ys, ref) < (1e-5 if variant == "pair" else 1e-2)
    if variant == "pair":
        assert rel_err(y2s, y2t) < 1e-2 and rel_err(st2s, st2t) < 1e-4
    # deterministic
    cs2 = run("slab")[0]
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("xf32", [True, False])
@pytest.mark.parametrize("N,D,K,ks", [(500, 256, 1024, 256), (1000 + 37, 512, 2048, 512), (333, 256, 512, 256)])
def test_layernorm_bwd_from_slabs_equals_the_two_launch_path(N, D, K, ks, xf32):
    """smx_layernorm_bwd2_slabs: the LayerNorm backward with its incoming gradient given as the split-K slabs of the dgrad behind it,
    against (tiled dgrad GEMM -> bf16 gradient -> smx_layernorm_bwd2): dX, the second output (alpha * D(dX) * mask, same keep
    decisions), the dgamma / dbeta partial rows.  The slab path never rounds the gradient to bf16, so it is compared at bf16 tolerance
    with the two-launch path and at float32 tolerance with float64 math."""
    g = torch.Generator(device="cuda").manual_seed(N + D)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1
    dz, W = rnd(N, K).bfloat16(), (rnd(K, D) * (2.0 / K ** 0.5)).bfloat16()          # dgrad: dh (N, D) = dz (N, K) W (K, D)
    x = rnd(N, D) * 2 if xf32 else (rnd(N, D) * 2).bfloat16()
    gam, bet = rnd(D) + 1.5, rnd(D) * 0.1
    st = torch.stack([x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    res = rnd(N, D).bfloat16()
    mask = (torch.rand(N, device="cuda", generator=g) > 0.3).to(torch.uint8)
    ns = K // ks
    slabs = torch.empty(ns, N, D, device="cuda")
    ops.gemm_panel_slabs(dz, ops.weight_pack_slices(W, ks, transposed=True), slabs, N, D, ks, ns)
    nb = L.lib().smx_layernorm_bwd_blocks(N)
    ws1, ws2 = torch.zeros(nb * 2 * D, device="cuda"), torch.zeros(nb * 2 * D, device="cuda")
    second = (0.5, mask, (0.15, 4321))
    dx1, dx1b = ops.layernorm_bwd(ops.Slabs(slabs), x, gam, bet, st, None, None, res=res, ws=ws1, second=second)
    dh = torch.empty(N, D, device="cuda", dtype=torch.bfloat16)
    ops.gemm(L.GEMM_NN, dz, W, dh, N, D, K)
    dx2, dx2b = ops.layernorm_bwd(dh, x, gam, bet, st, None, None, res=res, ws=ws2, second=second)
    assert rel_err(dx1, dx2) < 1.2e-2 and rel_err(dx1b, dx2b) < 1.2e-2
    assert float(((dx1b == 0) != (dx2b == 0)).float().mean()) < 2e-3                 # the same keep decisions (tiny values that round to 0 aside)
    assert rel_err(ws1.view(nb, 2, D).sum(0), ws2.view(nb, 2, D).sum(0)) < 1e-2
    # float64 reference of dX from the exact gradient
    gy = (dz.double() @ W.double()) * gam.double()
    xh = (x.double() - st[:, :1].double()) * st[:, 1:].double()
    ref = st[:, 1:].double() * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True)) + res.double()
    assert rel_err(dx1, ref) < 8e-3
    # bit-identical when repeated
    dx3, _ = ops.layernorm_bwd(ops.Slabs(slabs), x, gam, bet, st, None, None, res=res, ws=ws1, second=second)
    assert torch.equal(dx1, dx3)
