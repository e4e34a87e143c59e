"""Seeded random-shape sweeps of the round-6 small-batch kernels against float64 torch math: smx_pool_bcast, smx_gemm_panel_slabs +
smx_slab_epilogue (bias + residual form), smx_wgrad_group_direct.  The fixed cases live in test_kernels_gpu.py / test_splitk_gpu.py;
this file walks ragged sizes nobody chose by hand (one frame, one utterance, widths that are no multiple of a vector, panels shorter
than their height, frame counts around the 64-frame stage)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from summarymixing_amd import _lib as L, ops      # noqa: E402
from tests._util import rel_err                   # noqa: E402


def _cases(seed, n, gen):
    r = random.Random(seed)
    return [gen(r) for _ in range(n)]


@pytest.mark.parametrize("B,T,D,dtype", _cases(11, 24, lambda r: (r.choice([1, 2, 3, 5, 9, 16]), r.choice([1, 2, 7, 63, 64, 65, 300, 511, 1025]),
                                                                    r.choice([1, 5, 8, 24, 40, 64, 72, 200, 256, 264, 520]),
                                                                    r.choice([torch.float32, torch.bfloat16]))))
def test_pool_bcast_random_shapes(B, T, D, dtype):
    if not ops.pool_bcast_ok(B, T, D):
        pytest.skip("shape outside smx_pool_bcast_ok")
    g = torch.Generator(device="cuda").manual_seed(B * 7919 + T * 31 + D)
    s = (torch.rand(B * T, D, device="cuda", generator=g) * 2 - 1).to(dtype)
    mask = (torch.rand(B * T, device="cuda", generator=g) > 0.4).to(torch.uint8)
    mask.view(B, T)[:, -1] = 1
    ds = torch.full((B * T, D), 3.0, device="cuda", dtype=dtype)
    mean, inv = ops.pool_bcast(s, mask, B, T, ds=ds, scale=True, want_mean=True, want_inv=True)
    cnt = mask.view(B, T).sum(1, keepdim=True).double()
    ref = (s.double().view(B, T, D) * mask.view(B, T, 1)).sum(1) / cnt
    assert rel_err(mean, ref) < 3e-6
    assert rel_err(inv, 1.0 / cnt.view(-1)) < 1e-6
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert rel_err(ds.view(B, T, D), ref.view(B, 1, D).expand(B, T, D)) < tol     # the broadcast: every row of an utterance = its mean
    # the backward form: sum over ALL frames (no mask in), times inv_in, times the row mask
    d2 = torch.full((B * T, D), 3.0, device="cuda", dtype=dtype)
    ops.pool_bcast(s, None, B, T, ds=d2, scale=False, want_mean=False, inv_in=inv, mask_out=mask)
    ref2 = (s.double().view(B, T, D).sum(1) / cnt).view(B, 1, D) * mask.view(B, T, 1)
    assert rel_err(d2.view(B, T, D), ref2) < (3e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("N,M,ks,ns", _cases(12, 20, lambda r: (r.choice([1, 31, 32, 33, 64, 100, 257, 1000, 2047, 3750]), r.choice([64, 128, 192, 256, 320, 512]),
                                                                 r.choice([256, 512]), r.choice([1, 2, 3, 4, 8]))))
def test_slabs_and_reducer_random_shapes(N, M, ks, ns):
    K = ks * ns
    if not (L.lib().smx_gemm_panel_slabs_ok(L.BF16, N, M, ks, ns) == 1 and L.lib().smx_slab_epilogue_ok(L.BF16, N, M, ns) == 1):
        pytest.skip("shape outside smx_gemm_panel_slabs_ok / smx_slab_epilogue_ok")
    g = torch.Generator(device="cuda").manual_seed(N * 13 + M + ks + ns)
    x = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(M, K, device="cuda", generator=g) * 2 - 1) * (2.0 / K ** 0.5)).bfloat16()
    bias = torch.rand(M, device="cuda", generator=g) - 0.5
    res = torch.randn(N, M, device="cuda", generator=g)
    slabs = torch.full((ns, N, M), 7.0, device="cuda")
    ops.gemm_panel_slabs(x, ops.weight_pack_slices(W, ks), slabs, N, M, ks, ns)
    prod = x.double() @ W.double().t()
    assert rel_err(slabs.double().sum(0), prod) < 3e-6
    out = torch.full((N, M), 5.0, device="cuda")
    ops.slab_epilogue(slabs, ns, out, N, M, ops.epilogue(bias=bias, res=res, alpha=0.5, out_mode=L.OUT_F32))
    assert rel_err(out, res.double() + 0.5 * (prod + bias.double())) < 3e-6


@pytest.mark.parametrize("rows,shapes", _cases(13, 12, lambda r: (r.choice([1, 2, 63, 64, 65, 127, 128, 500, 1000, 3750, 4097]),
                                                                   [(128 * r.randint(1, 4), 128 * r.randint(1, 4)) for _ in range(r.randint(1, 6))])))
def test_wgrad_group_direct_random_shapes(rows, shapes):
    if any(L.lib().smx_wgrad_group_direct_ok(rows, M, K) != 1 for M, K in shapes):
        pytest.skip("shape outside smx_wgrad_group_direct_ok")
    torch.manual_seed(rows + len(shapes))
    recs, refs = [], []
    for i, (M, K) in enumerate(shapes):
        dz = (torch.randn(rows, M, device="cuda") * 0.5).bfloat16()
        x = torch.randn(rows, K, device="cuda").bfloat16()
        gW = torch.randn(M, K, device="cuda")
        gb = torch.randn(M, device="cuda") if i % 2 == 0 else None
        refs.append((gW.double() + dz.double().t() @ x.double(), None if gb is None else gb.double() + dz.double().sum(0)))
        recs.append((dz, x, gW, gb, M, K))
    ops.wgrad_group_direct(recs, rows)
    for (dz, x, gW, gb, M, K), (rw, rb) in zip(recs, refs):
        assert rel_err(gW, rw) < 3e-6, (M, K, rel_err(gW, rw))
        if gb is not None:
            assert rel_err(gb, rb) < 3e-6
