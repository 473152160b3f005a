import os
import sys

import pytest

# The parity tests drive single kernels and A/B knobs through include/fact_hip_debug.h: bind the test / bench build of the
# library (mint_amd/_lib.py).  Must be set before mint_amd._lib is imported.  tests/test_cabi.py checks the PRODUCTION
# library's exports separately, and tests/test_gpu_production_lib.py runs a train step on it in its own process.
os.environ.setdefault("FACT_DEBUG_ABI", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
