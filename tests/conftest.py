import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("HW_POISON_EMPTY") == "1":
        # torch.empty() returns NaN / max-int filled memory: any kernel that counts on a fresh allocation being zero (true
        # on a just-booted GPU, false once other work has used the memory) shows up as a failure on every box
        import torch

        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


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
