import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("operators")


@pytest.fixture(scope="session")
def engine():
    """Session-wide f64 engine on cuda:0.  No fallback: a missing library / device is an error."""
    from pyphysim_amd.engine import Engine
    eng = Engine(0, "f64")
    yield eng
    eng.close()
