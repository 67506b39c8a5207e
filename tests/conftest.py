import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(with_ref=True)
    return O


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through its C-ABI.  No fallback: a missing library or GPU is an error, not a skip."""
    from segalign_amd import engine as E
    E.lib()
    return E
