import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def pcx():
    """The product binding; loading fails loudly when libpcx_hip.so is not built."""
    import pothoscomms_amd as p
    p._lib.load()
    return p


@pytest.fixture(scope="session")
def dev(pcx):
    from pothoscomms_amd import device
    return device
