import os
import sys

import pytest

# The product fuses a d_model = 256 LayerNorm into its neighbouring GEMM only from 36 864 frames up (functional._LN_FUSE_MIN_ROWS:
# below that the 128-row LayerNorm tile underfills the chip).  The parity tests run at a few hundred to ~16 000 frames and are the
# encoder-level coverage of those fused epilogues, so they keep the fusion on at every size; tests/test_encoder_gpu.py compares
# the two paths directly.
os.environ.setdefault("SMX_LN_FUSE_MIN_ROWS", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
