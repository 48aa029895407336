import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: skip instead of failing in hipInit.
    if any("gpu" in it.keywords for it in items) and not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU visible")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)
