"""GPU: calls in the wrong order end with the reference's kind of error -- a message on stderr and exit code 1 -- never with a hang.
(An entry that needs a device slot used to wait for a token for ever when InitializeProcessor had not filled the pool.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = """
import sys, numpy as np
sys.path.insert(0, %r)
from segalign_amd import engine as E
E.InitializeInterface(1)
E.GenerateShapePos("TTT0T00TT00T0T0TTTT")
what = sys.argv[1]
if what == "coverage":
    E.RmCoverageIntervals(np.zeros(3, dtype=E.SEG_DTYPE), 1000, 1)
elif what == "range":
    E.SeedAndFilterRange(0, 1000, False, 0)
elif what == "rm":
    E.RmSeedAndFilter(np.zeros(4, dtype=np.uint64), False, 0, 100)
print("NOT REACHED")
"""


@pytest.mark.parametrize("what", ["coverage", "range", "rm"])
def test_a_call_before_initialize_processor_exits_with_code_1(what):
    out = subprocess.run([sys.executable, "-c", CODE % ROOT, what], capture_output=True, text=True, timeout=120)
    assert "NOT REACHED" not in out.stdout
    if what == "rm":   # (MAX_SEEDS is 0 before InitializeProcessor: the reference's own `assert(num_seeds <= MAX_SEEDS)` fires first, rm :726-730)
        assert out.returncode != 0
    else:
        assert out.returncode == 1, (out.returncode, out.stderr[-500:])
        assert "InitializeProcessor" in out.stderr
