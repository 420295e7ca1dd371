import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's convolutions are small (32 .. 224 pixel frames, a handful of them): on the GPU box's 256-CPU host
    # PyTorch's default of one thread per CPU makes them 10-40x SLOWER than 8-16 threads (bench.py's thread probe measures
    # 42.8 frames/s at 8 threads, 1.0 at 256) - that, not the GPU, was most of round 4's 1 076 s.
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


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
