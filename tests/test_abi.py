"""CPU-side checks of the drop-in boundary: libsmx.so loads and exports every symbol include/smx.h declares,
the ctypes table mirrors the header, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "smx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from summarymixing_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsmx.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/smx.h but not exported"


def test_ctypes_table_matches_header():
    from summarymixing_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_string_need_no_gpu():
    from summarymixing_amd import _lib
    L = _lib.lib()
    assert L.smx_version() == 100
    assert isinstance(L.smx_last_error(), bytes)


def test_epilogue_struct_layout_matches_header():
    from summarymixing_amd import _lib
    assert ctypes.sizeof(_lib.Epilogue) == 272  # 34 x 8 bytes (round 3: + io_flags, pad_, epoch), see include/smx.h smx_epilogue
    assert _lib.Epilogue.io_flags.offset == 256 and _lib.Epilogue.epoch.offset == 264


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    m = SummaryMixing(8, 1, [8], 8, [8], 8, global_dropout=0.0)
    with pytest.raises((AssertionError, RuntimeError)):
        m(torch.randn(1, 3, 8))


def test_wgrad_item_struct_layout_matches_header():
    from summarymixing_amd import _lib
    assert ctypes.sizeof(_lib.WgradItem) == 56  # 5 x 8 + 4 x 4 bytes, see include/smx.h smx_wgrad_item
    assert ctypes.sizeof(_lib.ReduceJob) == 56


def test_config_is_read_once_and_queryable():
    """smx_get_config returns the knobs the library read from the environment (include/smx.h: smx_config); a later change
    of the environment does not reach the library (no getenv on any call path)."""
    from summarymixing_amd import _lib
    before = _lib.get_config()
    assert before["epi_simple"] == int(os.environ.get("SMX_EPI_SIMPLE", 2)) and before["wgroup_bk"] in (32, 64)
    assert before["diag_build"] == 0 and before["gemm_ablate"] == 0 and before["wgroup_ablate"] == 0 and before["dwroll_ablate"] == 0
    old = os.environ.get("SMX_WGROUP_PP")
    os.environ["SMX_WGROUP_PP"] = "2" if before["wgroup_pp"] != 2 else "0"
    try:
        assert _lib.get_config() == before
    finally:
        if old is None:
            del os.environ["SMX_WGROUP_PP"]
        else:
            os.environ["SMX_WGROUP_PP"] = old
