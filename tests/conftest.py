import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """The built .so files are git-ignored: a fresh tree has none.  Build them once per session
    (hipcc cross-compiles without a GPU); the package itself never builds implicitly -- it fails
    loudly when the HIP library is missing."""
    import gym_acas2d_amd as g
    if not os.path.exists(g.native.LIB_PATH):
        g.native.build()
    yield


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O
