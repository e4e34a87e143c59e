"""Pin the CPU oracle (oracle/smx_oracle.py) against outputs of the reference itself.

The fixtures were produced by tests/golden/make_golden.py importing the unmodified reference
(SURVEY.md §8c).  fp32 tolerance 1e-5 abs / 1e-5 rel (same arithmetic, different op order)."""
import pytest
import torch

from oracle import smx_oracle as O
from tests import _golden as G

TOL = dict(rtol=1e-5, atol=2e-5)


def _cell(meta, arrays, sd, x):
    return O.summary_mixing(x, sd, "", meta["mode"], meta["act"], meta["local_proj_out_dim"],
                            arrays.get("sum_mask"), arrays.get("pad_mask"))


@pytest.mark.parametrize("name", G.names("g1_") + G.names("g2_"))
def test_cell_forward_and_grads(name):
    meta, arrays, sd, grads = G.load(name)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and k != "decay_constant") for k, v in sd.items()}
    x = arrays["x"].clone().requires_grad_(True)
    y = _cell(meta, arrays, sd, x)
    assert y.shape == arrays["y"].shape
    torch.testing.assert_close(y, arrays["y"], **TOL)
    (y * arrays["r"]).sum().backward()
    torch.testing.assert_close(x.grad, arrays["gx"], **TOL)
    for k, g in grads.items():
        torch.testing.assert_close(sd[k].grad, g, **TOL)


def test_parallel_linear():
    meta, a, sd, _ = G.load("g3_parallel_linear")
    torch.testing.assert_close(O.parallel_linear(a["x3"], sd["a.weights"], sd["a.biases"], True), a["y3"], **TOL)
    torch.testing.assert_close(O.parallel_linear(a["x4"], sd["b.weights"], sd["b.biases"], False),
                               a["y4_nocombine"], **TOL)
    torch.testing.assert_close(O.parallel_linear(a["x3"], sd["b.weights"], sd["b.biases"], False),
                               a["y3_nocombine"], **TOL)


@pytest.mark.parametrize("name", ["g5_conformer_layer_swish", "g5_conformer_layer_gelu", "g5_branchformer_layer"])
def test_encoder_layer(name):
    meta, arrays, sd, grads = G.load(name)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = arrays["x"].clone().requires_grad_(True)
    fn = O.conformer_layer if meta["kind"] == "conformer_layer" else O.branchformer_layer
    y = fn(x, sd, "", meta["act"], meta["mode"], meta["local_proj_out_dim"], None, arrays["pad_mask"])
    torch.testing.assert_close(y, arrays["y"], rtol=1e-4, atol=1e-4)
    (y * arrays["r"]).sum().backward()
    torch.testing.assert_close(x.grad, arrays["gx"], rtol=1e-4, atol=1e-4)
    for k, g in grads.items():
        torch.testing.assert_close(sd[k].grad, g, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["g5_config1_encoder", "g5_config1_encoder_dynchunk", "g5_branchformer_encoder"])
def test_asr_encode(name):
    meta, arrays, sd, _ = G.load(name)
    if not sd:  # the dynchunk fixture shares the weights of g5_config1_encoder
        sd = G.load("g5_config1_encoder")[2]
    dyn = tuple(meta["dynchunk"]) if meta["dynchunk"] else None
    y = O.asr_encode(arrays["src"], arrays["wav_len"], sd, meta["encoder_module"], meta["act"], meta["mode"],
                     meta["local_proj_out_dim"], dyn)
    torch.testing.assert_close(y, arrays["y"], rtol=1e-4, atol=1e-4)


def test_quirk_all_padding_row_is_nan():
    meta, arrays, sd, _ = G.load("g6_allpad_row")
    y = _cell(meta, arrays, sd, arrays["x"])
    assert torch.equal(torch.isnan(y), arrays["isnan"].bool())
    torch.testing.assert_close(y[0], arrays["y"][0], **TOL)


def test_quirk_padded_content_changes_valid_frames():
    meta, a, sd, _ = G.load("g6_padded_content")
    ya = O.conformer_layer(a["x"], sd, "", meta["act"], meta["mode"], meta["local_proj_out_dim"], None, a["pad_mask"])
    yb = O.conformer_layer(a["xb"], sd, "", meta["act"], meta["mode"], meta["local_proj_out_dim"], None, a["pad_mask"])
    torch.testing.assert_close(ya, a["y"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(yb, a["yb"], rtol=1e-4, atol=1e-4)
    valid = a["pad_mask"][1]
    assert (ya[1][valid] - yb[1][valid]).abs().max() > 1e-3  # padding is NOT neutral (Conformer.py:327-331)


def test_dynchunk_mask_matches_reference_builder():
    for name in G.names("g2_"):
        meta, a, _, _ = G.load(name)
        m = O.dynchunk_sum_mask(a["sum_mask"].shape[0], meta["chunk_size"], meta["left_context"])
        assert torch.equal(m, a["sum_mask"].bool())


def test_lite_mode_is_stride0_view_in_reference():
    meta, _, _, _ = G.load("g1_sm_lite_h1_nomask")
    assert meta["y_stride"][1] == 0  # summary_mixing.py:308 returns expand(); documented quirk
