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


def pytest_sessionstart(session):
    """The suites load libmcle.so; build it in-tree when a fresh checkout has none (hipcc cross-compiles gfx950
    without a GPU).  `__graft_entry__.build()` does the same."""
    lib = os.path.join(REPO, "pyphysim_amd", "csrc", "libmcle.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(REPO, "pyphysim_amd", "csrc"), "-j8"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


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
