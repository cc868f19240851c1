import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
ORACLE = os.path.join(REPO, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)
GOLDEN = os.path.join(REPO, "tests", "golden")
# compiled plans are not cached on disk under test (tests of the cache point it at a temp directory)
os.environ.setdefault("PYCHAIN_PLAN_CACHE_DIR", "off")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (runs on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
