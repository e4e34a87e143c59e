import os
import sys

import pytest

# The product fuses a d_model = 256 LayerNorm into its neighbouring GEMM only from 17 500 frames up (36 864 until round 6) (functional._LN_FUSE_MIN_ROWS:
# below that the 128-row LayerNorm tile underfills the chip).  The parity tests run at a few hundred to ~16 000 frames, so the
# encoder-level modules below run TWICE (fixture `ln_fuse_mode`, requested through their `pytestmark`):
#   lnfuse_default - the shipped dispatch (separate LayerNorm kernels at these sizes: what a user gets at the recipe batch),
#   lnfuse_always  - threshold 0, so the LayerNorm-fused GEMM epilogues are exercised against the same oracle values.
# Nothing is forced through the environment any more: every other test runs the shipped defaults.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_generate_tests(metafunc):
    if "ln_fuse_mode" in metafunc.fixturenames:
        metafunc.parametrize("ln_fuse_mode", ["lnfuse_default", "lnfuse_always"], indirect=True)


@pytest.fixture
def ln_fuse_mode(request):
    from summarymixing_amd import functional as F
    saved = F._LN_FUSE_MIN_ROWS
    F._LN_FUSE_MIN_ROWS = 0 if request.param == "lnfuse_always" else saved
    try:
        yield request.param
    finally:
        F._LN_FUSE_MIN_ROWS = saved
