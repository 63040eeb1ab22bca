import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI tests need hopwise_amd/csrc/libhopwise_amd.so; build it (hipcc cross-compiles gfx950 without a GPU) when a
    fresh checkout runs the tests before __graft_entry__.build()."""
    from hopwise_amd import build as hb

    if not os.path.exists(hb.LIB) and os.path.exists(hb.HIPCC):
        hb.build(verbose=False)
    if not os.path.exists(hb.TORCH_LIB) and os.path.exists(hb.HIPCC):
        hb.build_torch_ops(verbose=False)
