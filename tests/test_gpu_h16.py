"""GPU: hazard H16 -- heavy buckets against a small MAX_HITS: every planned iteration stops in front of the seed word that would take it
past MAX_HITS, the reserved num_hits / MAX_HITS + 2 iterations do not cover the call, and the LAST one takes what is left, more than MAX_HITS hits.
The reference overruns its buffers there; the oracle and the engine run that iteration at its real size.  This is the design that corrupted the heap of
the emulated reference (tests/golden/make_path_golden.py, 60 copies of a 37-base unit against MAX_HITS = 128): the engine must equal the oracle on it,
through the drop-in and the device-seeded entry, and the regime must really be reached."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from helpers import Case  # noqa: E402
from segalign_amd import shard  # noqa: E402

pytestmark = pytest.mark.gpu


def test_last_iteration_larger_than_max_hits(oracle, engine):
    from make_path_golden import design
    E, O = engine, oracle
    t, q = design(3, 7000, 4000, 60, 0.03)
    MAXH = 128
    try:
        E.reset_option(None)
        c = Case(t, q, chunk=800, transition=False).oracle_setup(O).engine_setup(E)
        E.set_max_hits(MAXH)
        over = checked = 0
        q_len = q.size - 19
        for rev in (False, True):
            for (a, b) in shard.chunks_of((0, q_len), 800, q_len, rev):
                seeds = c.host_seeds(a, b, rev)
                if seeds.size == 0:
                    continue
                want, st = c.oracle_saf(seeds, rev, max_hits=MAXH)
                # the plan on the oracle's scan: is the last iteration above MAX_HITS?
                counts = np.diff(np.concatenate([[0], c.o_index.astype(np.int64)]))[(seeds >> np.uint64(32)).astype(np.int64)]
                scan = np.cumsum(counts)
                n = int(scan[-1])
                if n >= MAXH:
                    start, limit = 0, MAXH
                    for _ in range(n // MAXH + 1):
                        pos = int(np.searchsorted(scan, limit, side="left")) - 1
                        if pos < 0:
                            break
                        start = int(scan[pos])
                        limit = min(start + MAXH, n)
                    over += (n - start) > MAXH
                for got in (E.SeedAndFilter(seeds, rev, 0), E.SeedAndFilterRange(a, b, rev, 0)):
                    assert np.array_equal(got, want), (rev, a, b)
                checked += 1
        assert checked >= 8 and over > 0
    finally:
        E.set_max_hits(0)
        E.ShutdownProcessor()
        E.reset_option(None)
