import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def ctx():
    """One vkv context on cuda:0 for the whole GPU session; fails loudly if the HIP library is missing."""
    import torch
    from vkvolume_amd import lib
    assert torch.cuda.is_available(), "GPU test started without a GPU"
    torch.cuda.set_device(0)
    c = lib.Context(0)
    yield c
    c.close()
