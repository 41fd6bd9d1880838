import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        if name.endswith(".json"):
            import json
            with open(os.path.join(GOLDEN, name)) as f:
                return json.load(f)
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max())
