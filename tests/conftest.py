import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))  # see oracle/oracle.py

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()  # builds liboracle.so on first use
    return O


@pytest.fixture(scope="session")
def gpu_lib():
    """libcup2d_hip.so on a machine with a GPU; fails (not skips) when the library is missing."""
    import cup2d_amd
    return cup2d_amd.load_library()
