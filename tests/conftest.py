import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (checker) and, when hipcc is present, the HIP library are built."""
    import oracle
    oracle.build()
    from nimblephysics_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__ as g
        g.build()
    yield
