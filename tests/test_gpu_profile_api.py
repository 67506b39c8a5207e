"""GPU: the measurement hooks bench.py builds its roofline block on -- per-scope HIP-event totals, the time a scope was RUNNING
(union of its launches over the slots' streams) and the call statistics (hits, hits forwarded by the class filter, lookup path)."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def test_busy_time_and_call_statistics(oracle, engine):
    t, q = synth.make_pair(3_000_000, 51, 52, sub_rate=0.08, mask_frac=0.1, records=2)
    c = Case(t, q[:1_000_000], chunk=250_000).oracle_setup(oracle).engine_setup(engine)
    E = engine
    try:
        assert E.lookup_mode() == 2
        calls = [(0, 1_000_000 - 19, False), (0, 1_000_000 - 19, True)] * 4
        E.profile_reset()
        E.profile_enable(True)
        outs, st = E.SeedCalls(calls, 0, 4)  # four calls in flight: the launches of a scope overlap
        E.profile_enable(False)
        prof = E.profile_entries()
        assert "extend_filter" in prof and "seed_probe" in prof
        total_ms, launches = prof["extend_filter"]
        assert launches >= len(calls)  # (a call whose hits split into two reference iterations launches the filter twice)
        busy = E.profile_busy_ms("extend_filter")
        # the union of the launches is never longer than their sum, and never shorter than the longest one
        assert 0.0 < busy <= total_ms * 1.0001
        assert busy >= total_ms / launches * 0.999
        assert E.profile_busy_ms("no such scope") == 0.0
        # statistics: the class filter forwards a few per cent of the hits, far more than survive the second level
        assert st["lookup_path"] == 2
        assert 0 < st["num_forwarded"] < 0.2 * st["num_hits"]
        assert st["num_candidates"] <= st["num_forwarded"]
        # and the calls are what they are one by one (chunks of a call concatenated, headers removed)
        for (a, b, rev), o in zip(calls[:2], outs[:2]):
            want = [c.oracle_saf(c.host_seeds(s, e, rev), rev)[0][1:] for (s, e) in c.chunks()]
            assert seg_equal(o, np.concatenate(want))
    finally:
        E.ShutdownProcessor()
