import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both_gemm_modes: run under the whole-tile AND the tuned GEMM dispatch")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_generate_tests(metafunc):
    """Tests marked ``both_gemm_modes`` (the tolerance-based comparisons of the model against the reference's fixtures) run
    twice: under the whole-tile GEMM dispatch the GPU suite pins, and under the TUNED dispatch production and bench.py use
    (shipped decision table, online tuner, split-K pairs, packed weights, fused RoPE) — VERDICT r3 item 8."""
    if metafunc.definition.get_closest_marker("both_gemm_modes") is not None:
        metafunc.parametrize("_gemm_mode", ["tiles", "tuned"], indirect=True)


@pytest.fixture
def _gemm_mode(request):
    return getattr(request, "param", "tiles")


@pytest.fixture(autouse=True)
def _deterministic_gemm_dispatch(request, monkeypatch, _gemm_mode):
    """GPU tests compare runs with each other: pin the GEMM dispatch to the whole-tile kernels (bit-identical
    across calls and batch sizes).  Tests of the tuned dispatch switch the mode themselves, or carry the ``both_gemm_modes``
    marker (then this fixture sets the mode of the run)."""
    if "gpu" in request.keywords:
        from valley_amd import ops
        # VALLEY_TEST_GEMM_MODE=tuned: the whole suite under the production dispatch (an exploratory run: tests that compare
        # two runs bit for bit may then see two different kernels of the online tuner)
        monkeypatch.setattr(ops, "GEMM_MODE", os.environ.get("VALLEY_TEST_GEMM_MODE", _gemm_mode))
    yield
