import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library; GPU tests fail loudly (never skip to a fallback) if missing"""
    import torch
    assert torch.cuda.is_available(), "GPU test without a GPU"
    from bwtb3m_amd import _lib
    _lib.lib()
    return torch
