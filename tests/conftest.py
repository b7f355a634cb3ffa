import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built here if stale; loads without a GPU)."""
    from pioneer_amd import _lib
    _lib.build_library()
    return _lib.load_library()


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import build_oracle
    return build_oracle()
