import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def tiled_gemm_only():
    """Bitwise batch-invariance tests: an embedding is bit-identical whatever else shares its batch as long as the same GEMM kernel family
    runs it.  Calls of <= 80 rows take the column-sliced skinny kernels (csrc/gemm_small.hip: a different, fixed k-summation order), so
    tests that compare a tiny call with a large one bit for bit pin the tiled family; tests/test_small_m_gpu.py bounds the difference."""
    from marqo_amd import _lib as L
    lib = L.load()
    L.check(lib.mq_tune(b"small_m", 0))
    yield
    L.check(lib.mq_tune(b"small_m", 80))
