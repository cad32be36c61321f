import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def ulp_diff(a, b):
    """Distance in units in the last place between two float64 arrays."""
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return np.abs(a - b)


def rel_to_max(a, b):
    """max|a-b| / max|b| -- the profile metric of SURVEY.md section 7 'Hard parts' 1."""
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))
