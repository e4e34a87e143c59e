"""GPU parity of the HIP SummaryMixing cell against the reference-generated golden fixtures and the oracle.
Tolerances (north_star): forward 1e-3 rel fp32 / 1e-2 bf16; gradients 1e-3 fp32 / 3e-2 bf16 (max-abs error
relative to the max-abs of the reference tensor)."""
import pytest
import torch

from tests import _golden as G
from tests._util import TOL, cell_dims_from_sd, rel_err

pytestmark = pytest.mark.gpu

ACT = {"gelu": torch.nn.GELU, "swish": "swish", "leaky_relu": torch.nn.LeakyReLU}


def _build(meta, sd, enc_dim):
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    kw = cell_dims_from_sd(sd, meta, enc_dim)
    m = SummaryMixing(activation=ACT[meta["act"]], global_dropout=0.0, **kw)
    missing = m.load_state_dict(sd, strict=True)
    return m.cuda()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", G.names("g1_") + G.names("g2_"))
def test_cell_matches_reference_golden(name, dtype):
    from summarymixing_amd import functional as F
    meta, a, sd, grads = G.load(name)
    m = _build(meta, sd, a["x"].shape[-1])
    x = a["x"].cuda().to(dtype).requires_grad_(True)
    pad = a["pad_mask"].cuda() if "pad_mask" in a else None
    sm = None
    if "sum_mask" in a:
        sm = F.DynChunkMask(a["sum_mask"].shape[0], meta["chunk_size"], meta["left_context"])
        assert torch.equal(sm.dense(), a["sum_mask"].bool())
    y = m(x, sum_mask=sm, src_padding_mask=pad)
    ftol, gtol = TOL[dtype]
    assert y.shape == a["y"].shape and y.dtype == dtype
    assert rel_err(y, a["y"]) <= ftol, f"forward rel err {rel_err(y, a['y'])}"
    (y.float() * a["r"].cuda()).sum().backward()
    assert rel_err(x.grad, a["gx"]) <= gtol, f"dx rel err {rel_err(x.grad, a['gx'])}"
    params = dict(m.named_parameters())
    for k, g in grads.items():
        e = rel_err(params[k].grad, g)
        assert e <= gtol, f"grad {k} rel err {e}"


@pytest.mark.parametrize("name", G.names("g2_"))
def test_dense_sum_mask_path_matches_chunk_path(name):
    """A dense (T,T) tensor sum_mask (what the reference passes) goes through the batched-GEMM path."""
    meta, a, sd, grads = G.load(name)
    m = _build(meta, sd, a["x"].shape[-1])
    x = a["x"].cuda().requires_grad_(True)
    y = m(x, sum_mask=a["sum_mask"].cuda(), src_padding_mask=a["pad_mask"].cuda())
    assert rel_err(y, a["y"]) <= 1e-3
    (y * a["r"].cuda()).sum().backward()
    assert rel_err(x.grad, a["gx"]) <= 1e-3


def test_all_padding_row_gives_nan_like_reference():
    meta, a, sd, _ = G.load("g6_allpad_row")
    m = _build(meta, sd, a["x"].shape[-1])
    y = m(a["x"].cuda(), src_padding_mask=a["pad_mask"].cuda())
    assert torch.equal(torch.isnan(y).cpu(), a["isnan"].bool())
    assert rel_err(y[0], a["y"][0]) <= 1e-3


def test_lite_returns_stride0_view_like_reference():
    meta, a, sd, _ = G.load("g1_sm_lite_h1_nomask")
    m = _build(meta, sd, a["x"].shape[-1])
    y = m(a["x"].cuda())
    assert y.stride(1) == 0


def test_bad_mode_raises_value_error():
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    with pytest.raises(ValueError):
        SummaryMixing(8, 1, mode="nope")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cell_fast_vs_oracle_random_shapes(dtype):
    """Seeded random inputs at a larger, ragged shape; oracle evaluated on the same weights in fp64."""
    from oracle import smx_oracle as O
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    torch.manual_seed(0)
    B, T, d, l = 5, 203, 144, 144
    m = SummaryMixing(d, 4, [d], l, [d], d, activation="swish", global_dropout=0.0, mode="SummaryMixing-fast")
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.08)
    sd = {k: v.double() for k, v in m.state_dict().items()}
    x = torch.randn(B, T, d)
    lens = torch.tensor([T, 17, 150, 1, 99])
    pad = torch.arange(T)[None] < lens[:, None]
    ref = O.summary_mixing(x.double(), sd, "", "SummaryMixing-fast", "swish", l, None, pad)
    y = m.cuda()(x.cuda().to(dtype), src_padding_mask=pad.cuda())
    assert rel_err(y, ref) <= TOL[dtype][0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_expdecay_cell_uses_the_linear_time_kernels(dtype, monkeypatch):
    """SummaryMixing-expdecay without sum_mask runs on the O(T) recurrence kernels and still matches
    the oracle's dense (T,T) Laplace evaluation, forward and backward, at a ragged mid-size shape."""
    from oracle import smx_oracle as O
    from summarymixing_amd import ops
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    torch.manual_seed(4)
    B, T, d = 3, 301, 64
    m = SummaryMixing(d, 1, [d], d, [d], d, activation="gelu", global_dropout=0.0, mode="SummaryMixing-expdecay")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.requires_grad:
                p.normal_(0, 0.1)
    sd = {k: v.double().clone().requires_grad_(v.requires_grad) for k, v in m.state_dict().items()}
    for k, p in m.named_parameters():
        sd[k].requires_grad_(p.requires_grad)
    x = torch.randn(B, T, d)
    pad = torch.arange(T)[None] < torch.tensor([T, 120, 250])[:, None]
    r = torch.randn(B, T, d)
    xo = x.double().requires_grad_(True)
    ref = O.summary_mixing(xo, sd, "", "SummaryMixing-expdecay", "gelu", d, None, pad)
    (ref * r.double()).sum().backward()
    calls = []
    real = ops.expdecay_mean
    monkeypatch.setattr(ops, "expdecay_mean", lambda *a, **k: (calls.append(k.get("reverse", False)), real(*a, **k))[1])
    xg = x.cuda().to(dtype).requires_grad_(True)
    y = m.cuda()(xg, src_padding_mask=pad.cuda())
    (y.float() * r.cuda()).sum().backward()
    assert calls == [False, True]                      # one forward filter, one transposed filter in the backward
    assert rel_err(y, ref) <= TOL[dtype][0]
    assert rel_err(xg.grad, xo.grad) <= TOL[dtype][1]


@pytest.mark.parametrize("B,T", [(1, 1), (2, 3), (1, 130), (7, 2)])
@pytest.mark.parametrize("mode", ["SummaryMixing", "SummaryMixing-fast", "SummaryMixing-lite"])
def test_cell_degenerate_shapes_vs_oracle(B, T, mode):
    """Tiny and odd shapes (single frame, single utterance, T < chunk sizes of every kernel), no padding mask."""
    from oracle import smx_oracle as O
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    torch.manual_seed(B * 100 + T)
    d = 32
    m = SummaryMixing(d, 2, [d], d, [d], d, activation="gelu", global_dropout=0.0, mode=mode)
    sd = {k: v.double() for k, v in m.state_dict().items()}
    x = torch.randn(B, T, d)
    ref = O.summary_mixing(x.double(), sd, "", mode, "gelu", d, None, None)
    xg = x.cuda().requires_grad_(True)
    y = m.cuda()(xg)
    assert y.shape == ref.shape and rel_err(y, ref) <= 1e-3
    y.sum().backward()
    assert torch.isfinite(xg.grad).all()
