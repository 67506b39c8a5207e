"""CPU: the host seeding loop as the oracle and segalign_amd/shard.py restate it, against every g_SeedAndFilter call the reference's own
seeder_body::operator() makes (src/seeder.cpp compiled as it lies + the real common/ntcoding.cpp, TBB's tuple header stood in for:
tests/golden/make_seeder_golden.py): which chunks are called on which strand in which order (seeder.cpp:47-121; the minus strand walks
[q_len - end, q_len - start) of the reverse-complement buffer, :33-34,:89-91), chunks without a seed word skipped (:76), the seed words
themselves (k-mer word, then one per transition position in ascending t, :60-69), blocks that do not start at offset 0 of the DRAM arena,
12of19 and 14of22, transitions on and off, IUPAC letters in the query (hazard H14: the real RevComp shifts the minus strand).
A second route, not a pin."""
import numpy as np
import pytest

import seeder_golden as G
from segalign_amd import shard

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_host_loop_restatement_makes_the_reference_seeders_calls(oracle, c):
    O = oracle
    k = O.generate_shape_pos(c["shape"])
    span, qs, n, q_len = len(c["shape"]), c["q_block_start"], c["block_len"], c["q_len"]
    assert q_len == n - span                                             # src/main.cpp:708
    arena = c["arena"].tobytes()
    rc_block = O.rev_comp_ascii(arena, qs, n)                            # src/main.cpp:377 (the oracle's RevComp is pinned to the real one)
    want = []
    for kk, (s, e) in enumerate(c["intervals"]):
        for rev in (False, True):
            if not (c["strand"] & (2 if rev else 1)):
                continue
            for (a, b) in shard.chunks_of((s, e), c["chunk"], q_len, rev):
                seeds = O.make_seeds(rc_block, 0, a, b, span, k, bool(c["transition"])) if rev else O.make_seeds(arena, qs, a, b, span, k, bool(c["transition"]))
                if seeds.size:                                           # seeder.cpp:76 / :111
                    want.append((kk, int(rev), kk & 1, seeds))
    got = c["calls"]
    assert len(got) == len(want)
    for g, (kk, rev, buf, seeds) in zip(got, want):
        assert (g["interval"], g["rev"], g["buffer"], g["n"]) == (kk, rev, buf, seeds.size)
        assert np.array_equal(g["seeds"], seeds)
    if c["transition"]:
        per = 1 + sum(1 for t in range(k) if O.is_transition_at_pos(t))
        assert all(g["n"] % per == 0 for g in got)


def test_the_golden_set_reaches_the_loops_corners():
    assert any(c["q_block_start"] for c in CASES) and {c["strand"] for c in CASES} == {1, 2, 3}
    assert any(c["iupac"] for c in CASES) and any(not c["iupac"] and c["strand"] & 2 for c in CASES)
    for c in CASES:   # a chunk of N: fewer calls than chunks
        chunks = sum(len(shard.chunks_of(tuple(iv), c["chunk"], c["q_len"], rev)) for iv in c["intervals"] for rev in (False, True) if c["strand"] & (2 if rev else 1))
        assert len(c["calls"]) <= chunks
    assert any(len(c["calls"]) < sum(len(shard.chunks_of(tuple(iv), c["chunk"], c["q_len"], rev)) for iv in c["intervals"] for rev in (False, True)
                                     if c["strand"] & (2 if rev else 1)) for c in CASES)
