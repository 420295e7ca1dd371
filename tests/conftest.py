import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The loaded C-ABI library (built in-tree by __graft_entry__.build())."""
    import orbit_dataset_amd  # noqa: F401
    from orbit_dataset_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)
