import os
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pydrobert-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    skip = None
    for item in items:
        if "gpu" in item.keywords:
            if skip is None:
                skip = (
                    pytest.mark.skip(reason="no ROCm device visible") if not _has_gpu() else False
                )
            if skip:
                item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed(request):
    """Per-test deterministic seeding (same idea as the reference's tests/conftest.py:88-89)."""
    import numpy as np
    import torch

    seed = zlib.adler32(request.node.name.encode("utf-8"))
    torch.manual_seed(seed)
    np.random.seed(seed % (2**32))
    yield


@pytest.fixture
def device():
    import torch

    return torch.device("cuda:0")


@pytest.fixture
def switch():
    """``switch("PDT_BEAM_FUSED", 0)``: set one of the package's run-time switches (read from the
    environment once at import; pydrobert_amd/switches.py) for the rest of the test."""
    from pydrobert_amd import switches

    old = {}

    def set_(name, value):
        old.setdefault(name, switches.get(name))
        switches.set(name, int(value))

    yield set_
    for name, value in old.items():
        switches.set(name, value)
