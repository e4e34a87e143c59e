"""INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would add to
speechbrain/nnet/summary_mixing.py.  These tests EXECUTE that code block verbatim (extracted from the markdown), so the
document cannot drift from include/smx.h: the structure it declares must have the library's size, and - on the GPU - its
forward_mixing_fast must reproduce the package's own SummaryMixing-fast cell."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C-ABI binding"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the stub loads "summarymixing_amd/libsmx.so" relative to the checkout
    try:
        exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_stub_structure_matches_the_library():
    from summarymixing_amd import _lib
    ns = _stub_namespace()
    assert ctypes.sizeof(ns["Epi"]) == ctypes.sizeof(_lib.Epilogue) == 320
    for name in ("bias", "c0", "ldc0", "c0_mode", "c0_div", "act", "out_mode", "row_mask", "alpha", "flags", "io_flags", "epoch"):
        assert getattr(ns["Epi"], name).offset == getattr(_lib.Epilogue, name).offset, name
    assert (ns["SMX_BF16"], ns["SWISH"], ns["C0_GROUP"], ns["OUT_F32"]) == (_lib.BF16, _lib.ACT_SWISH, _lib.C0_GROUP, _lib.OUT_F32)


@pytest.mark.gpu
def test_stub_forward_matches_reference_golden():
    """The stub's forward_mixing_fast on the reference-generated fixture g1_sm_fast_h1_mask (tests/golden/make_golden.py):
    the reference module's own weights, input, padding mask and output."""
    from tests import _golden as G
    from tests.test_cell_gpu import _build
    ns = _stub_namespace()
    meta, a, sd, _ = G.load("g1_sm_fast_h1_mask")
    cell = _build(meta, sd, a["x"].shape[-1]).eval()        # holds the reference state_dict under the reference's attribute names
    x = a["x"].cuda().bfloat16()
    valid = a["pad_mask"].cuda()
    B, T, _ = x.shape
    got = ns["forward_mixing_fast"](cell, x, valid.to(torch.uint8).reshape(-1).contiguous()).float()
    torch.cuda.synchronize()
    ref = a["y"].cuda()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert got.shape == ref.shape and err <= 1e-2, err       # north_star bf16 forward tolerance
    with torch.no_grad():
        own = cell(x, src_padding_mask=valid).float()        # and the package's own cell on the same input
    assert (got - own).abs().max().item() / own.abs().max().item() <= 1e-2
