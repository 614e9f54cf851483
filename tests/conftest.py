import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a machine without a GPU: fail loudly rather than silently pass on a fallback.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    skip = pytest.mark.skip(reason="needs a GPU; run with -m gpu on the GPU box")
    if not _gpu_available():
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _finalizers_between_tests(request):
    """Objects caught in reference cycles (the traceback of an expected error keeps its frame's locals, an index among
    them) are finalized here, between tests and with the GPU idle, not wherever the cyclic collector happens to run
    inside a later test."""
    yield
    if "gpu" in request.keywords:
        import gc

        gc.collect()
