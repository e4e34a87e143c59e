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


def test_library_exports_nothing_undeclared():
    """The product library's dynamic `smx_*` symbols are exactly the header's: no debug hooks, no undeclared entry points
    (the clock-stamp setters live in the -DSMX_DIAG build only)."""
    import subprocess
    from summarymixing_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("smx_")})
    assert exported == _declared(), sorted(set(exported) ^ set(_declared()))


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
    assert ctypes.sizeof(_lib.Epilogue) == 320  # 40 x 8 bytes (round 3: + io_flags, pad_, epoch; round 5: + lnf2_*), see include/smx.h smx_epilogue
    assert _lib.Epilogue.io_flags.offset == 256 and _lib.Epilogue.epoch.offset == 264 and _lib.Epilogue.lnf2_eps.offset == 312


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
    assert before["t256"] == int(os.environ.get("SMX_T256", 1)) and before["ln_tile_rows"] == 128
    assert before["diag_build"] == 0 and before["gemm_ablate"] == 0 and before["wgroup_ablate"] == 0 and before["dwroll_ablate"] == 0
    old = os.environ.get("SMX_T256")
    os.environ["SMX_T256"] = "2" if before["t256"] != 2 else "0"
    try:
        assert _lib.get_config() == before
    finally:
        if old is None:
            del os.environ["SMX_T256"]
        else:
            os.environ["SMX_T256"] = old


def test_grouped_wgrad_slice_plan_is_size_aware():
    """smx_wgrad_group_splits is host logic (no GPU): the slice count comes from a cost model - rounds x 64-frame steps + one
    written-and-re-read fp32 slab per tile and slice.  At 64 000 frames it keeps the counts tuned in round 2 (C2b layer: 11,
    C2a layer: 8); at the recipe batch (3750 frames) a slice costs a sixth of the kernel and two beat the five that 'fill the
    chip'; a ragged frame count plans like its multiple of 64."""
    from summarymixing_amd import _lib as L
    lib = L.lib()

    def plan(rows, shapes):
        items = (L.WgradItem * len(shapes))()
        for it, (M, K) in zip(items, shapes):
            it.M, it.K, it.want_bias = M, K, 1
        return lib.smx_wgrad_group_splits(rows, items, len(shapes))
    c2b = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)]
    c2a = [(2048, 512), (512, 2048), (2048, 512), (512, 2048), (1024, 512), (512, 1024), (1024, 512), (512, 512)]
    assert plan(64000, c2b) == 11
    assert plan(64000, c2a) == 8
    assert plan(3750, c2a) == 2
    assert plan(3750, c2a) == plan(3712, c2a)
    assert plan(63, c2b) == 0 and plan(64, c2b) == 1          # fewer than 64 frames: not this kernel's job
    for rows in (64, 700, 3750, 16000, 64000, 240000):
        s = plan(rows, c2a)
        assert 1 <= s <= max(1, rows // 64 // 8) or s == 1    # at least 8 steps of 64 frames per slice


def test_launch_geometry_rules_need_no_gpu_and_hold_their_measured_cells():
    """The library's geometry choices are host arithmetic (no GPU): the panel height (smx_gemm_panel_rows) and the row-complete
    LayerNorm tile (smx_gemm_ln_tile_rows_for) at the sizes whose A/B tables are in tools/experiments/r06_smalln/README.md."""
    import subprocess
    import sys
    # (a fresh process with the knobs unset: the library reads SMX_PANEL_ROWS / SMX_LN_TILE64 once)
    code = ("import ctypes;from summarymixing_amd import _lib;L=_lib.lib();"
            "print([L.smx_gemm_panel_rows(n,m) for n,m in ((3750,2048),(3750,1024),(500,1024),(12000,512),(32000,1024),(36000,1024),(64000,1024),(240000,2048),(240000,512))]);"
            "print([L.smx_gemm_ln_tile_rows_for(n,m) for n,m in ((17500,256),(32000,256),(36000,256),(64000,256),(32000,512),(0,256))]);"
            "print([L.smx_pool_bcast_ok(b,t,d) for b,t,d in ((10,375,512),(1,500,256),(128,500,256),(8,30000,512),(2,4096,64),(2,4097,64))])")
    env = {k: v for k, v in os.environ.items() if k not in ("SMX_PANEL_ROWS", "SMX_LN_TILE64", "SMX_POOL_FUSE_MAX_ROWS")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, cwd=ROOT, env=env).stdout.splitlines()
    rows, tiles, pools = eval(out[-3]), eval(out[-2]), eval(out[-1])
    # 3750 frames: 64-row panels for the 2048-wide up-projection, 32 rows at 1024 (one round of 188-256 workgroups); one utterance: 32 rows;
    # 12 000 x 512: 64 rows; 32 000 rows = 250 panels of 128; 36 000 rows (282 panels) takes the geometry that fills its rounds; config 5's
    # 1875 panels stay on 128 rows (the round-efficiency rule stops at four rounds)
    assert rows[0] == 64 and rows[1] == 32 and rows[2] == 32 and rows[3] == 64 and rows[4] == 128 and rows[6] == 128
    assert rows[5] in (64, 128) and rows[7] == 128 and rows[8] == 128
    # the 64-row LayerNorm tile where 128-row tiles leave one workgroup per CU (<= 32 768 rows at d_model 256), never at d_model 512
    assert tiles == [64, 64, 128, 128, 128, 128]
    assert pools == [1, 1, 1, 0, 1, 0]
