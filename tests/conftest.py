import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU: they are deselected by
    `-m "not gpu"`; if someone runs them anyway without a device they FAIL in the fixture."""
    return


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a visible HIP device"
    from cadm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")
