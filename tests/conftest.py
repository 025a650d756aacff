import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the CPU oracle once (seconds); the HIP library is built by __graft_entry__.build()."""
    from oracle import oracle
    oracle.build()


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """torch and the HIP library share one HIP runtime in this process.  Tests that hand torch device buffers to the C-ABI or read
    torch.cuda.mem_get_info() need torch's lazy device initialisation to succeed whatever ran before them (it fails with "No HIP
    GPUs are available" when it first happens after some of the negative tests), so it is done once, up front, when a GPU is there."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
