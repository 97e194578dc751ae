import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gym-duckietown_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        # a wedged kernel must not hold the GPU box until the harness kills it: bound every GPU test
        # (the whole `-m gpu` suite takes ~75 s; pytest-timeout's thread method ends the process)
        if config.pluginmanager.hasplugin("timeout"):
            for it in items:
                if "gpu" in it.keywords and it.get_closest_marker("timeout") is None:
                    it.add_marker(pytest.mark.timeout(300, method="thread"))
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
