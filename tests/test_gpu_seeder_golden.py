"""GPU: the device seeder (sa_device_make_seeds: seeds.hip, what the device-seeded entries fall back to and what the drop-in check compares
against) emits, word for word, what the reference's own seeder_body::operator() hands g_SeedAndFilter (tests/golden/seeder_golden.json:
src/seeder.cpp compiled as it lies + the real ntcoding.cpp) -- plus strand always; minus strand where the query holds only ACGTacgtNn
(the engine mirrors the block on the device, the reference's host RevComp shifts it behind other letters: hazard H14)."""
import numpy as np
import pytest

import seeder_golden as G
from segalign_amd import shard

pytestmark = pytest.mark.gpu

CASES = list(G.cases())


def test_device_seeder_emits_the_reference_seeders_words(oracle, engine):
    E = engine
    checked = {False: 0, True: 0}
    try:
        for c in CASES:
            span, qs, n, q_len = len(c["shape"]), c["q_block_start"], c["block_len"], c["q_len"]
            E.reset_option(None)
            E.InitializeInterface(1)
            k = E.GenerateShapePos(c["shape"])
            assert oracle.generate_shape_pos(c["shape"]) == k   # (also: the oracle's transition positions, for the words per position below)
            E.InitializeProcessor(bool(c["transition"]), c["chunk"], span, oracle.build_sub_mat(910), 910, 3000, False)
            E.SendQueryWriteRequest(c["arena"], qs, n, 0)     # the block sits at q_block_start of the host arena (src/main.cpp:661)
            per = 1 + (sum(1 for t in range(k) if oracle.is_transition_at_pos(t)) if c["transition"] else 0)
            calls = iter(c["calls"])
            for kk, (s, e) in enumerate(c["intervals"]):
                for rev in (False, True):
                    if not (c["strand"] & (2 if rev else 1)):
                        continue
                    for (a, b) in shard.chunks_of((s, e), c["chunk"], q_len, rev):
                        got = E.device_make_seeds(a, b, rev, 0, per)
                        if got.size == 0:
                            continue                          # (no seed words: the reference makes no call, seeder.cpp:76)
                        g = next(calls)
                        assert (g["interval"], g["rev"]) == (kk, int(rev))
                        if rev and c["iupac"]:
                            continue                          # H14: the reference's minus-strand buffer is shifted behind the IUPAC letter
                        assert got.size == g["n"], (G.case_id(c), kk, rev, a, b, got.size, g["n"])
                        bad = np.nonzero(got != g["seeds"])[0]
                        assert bad.size == 0, (G.case_id(c), kk, rev, a, b, bad[:4].tolist(), [hex(int(x)) for x in got[bad[:2]]], [hex(int(x)) for x in g["seeds"][bad[:2]]])
                        checked[rev] += 1
            if not c["iupac"]:
                assert next(calls, None) is None
            E.ShutdownProcessor()
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)
    assert checked[False] > 15 and checked[True] > 5
