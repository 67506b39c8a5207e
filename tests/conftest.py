import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the test process is the engine's host: a hardware queue per slot
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


@pytest.fixture(scope="session")
def standin_100mbp():
    """BASELINE configs[1] stand-in (100 Mbp 7-record target x 8 %-diverged soft-masked query with inversions), generated
    once per session and shared by the full-size modules (configs[1], [3], [4])."""
    from segalign_amd import synth
    return synth.make_pair(100_000_000, 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, invert_frac=0.3,
                           invert_block=100_000)
