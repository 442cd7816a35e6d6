import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _deterministic_gemm_dispatch(request, monkeypatch):
    """GPU tests compare runs with each other: pin the GEMM dispatch to the whole-tile kernels (bit-identical
    across calls and batch sizes).  Tests of the tuned dispatch switch the mode themselves."""
    if "gpu" in request.keywords:
        from valley_amd import ops
        monkeypatch.setattr(ops, "GEMM_MODE", "tiles")
    yield
