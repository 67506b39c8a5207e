"""GPU: the engine against what the reference's OWN FILES return when they run end to end (tests/golden/path_golden.json, generator
tests/golden/make_path_golden.py): every g_SeedAndFilter call of the run -- header and HSPs in the reference's order -- through the drop-in
entry (the seed vector of the host loop), through the device-seeded entry (table-direct lookup) and through the grouped entry
(sa_seed_calls: an interval's chunks in one pass), with MAX_HITS set to what the reference's
arithmetic gives the generator's small "GPU" (sa_set_max_hits: calls in several iterations).  A second route, not a pin."""
import numpy as np
import pytest

import path_golden as G
from segalign_amd import shard

pytestmark = pytest.mark.gpu

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_engine_returns_what_the_reference_files_return(oracle, engine, c):
    E, O = engine, oracle
    span, ts, tl, qs, ql = len(c["shape"]), c["t_start"], c["t_len"], c["q_start"], c["q_len"]
    q_arena = c["query_arena"].tobytes()
    rc_block = O.rev_comp_ascii(q_arena, qs, ql)
    try:
        E.reset_option(None)
        E.InitializeInterface(1)
        k = E.GenerateShapePos(c["shape"])
        assert O.generate_shape_pos(c["shape"]) == k                                  # (the oracle's shape state: its seeder makes the drop-in's vectors)
        E.InitializeProcessor(bool(c["transition"]), c["chunk"], span, c["sub_mat"], c["xdrop"], c["hspthresh"], bool(c["noentropy"]))
        assert E.max_hits_for_mem(c["total_global_mem"]) == c["max_hits"]            # seed_filter.cu:833-841 on the generator's "GPU"
        E.set_max_hits(c["max_hits"])
        keep = E.SendRefWriteRequest(c["target_arena"], ts, tl)
        E.GenerateSeedPosTable(keep, ts, tl, c["step"], span, k)
        E.SendQueryWriteRequest(c["query_arena"], qs, ql, 0)
        calls = iter(c["calls"])
        for kk, rev, a, b in G.chunk_calls(c, shard):
            seeds = O.make_seeds(rc_block, 0, a, b, span, k, bool(c["transition"])) if rev else O.make_seeds(q_arena, qs, a, b, span, k, bool(c["transition"]))
            if seeds.size == 0:
                assert E.SeedAndFilterRange(a, b, rev, 0).size == 0
                continue
            g = next(calls)
            for name, got in (("drop-in", E.SeedAndFilter(seeds, rev, 0)), ("device-seeded", E.SeedAndFilterRange(a, b, rev, 0))):
                where = (G.case_id(c), name, kk, rev, a, b)
                assert (int(got[0]["len"]), int(got[0]["score"])) == (g["n_hsps"], g["num_hits"]), where
                assert np.array_equal(got[1:], g["hsps"]), where
        assert next(calls, None) is None
        # the grouped entry the bench and the hosts use: one call per interval and strand, its wga_chunk pieces sharing one pass over the kernels
        q_blk = ql - span
        for kk, (s, e) in enumerate(c["intervals"]):
            for rev in (False, True):
                if not (c["strand"] & (2 if rev else 1)):
                    continue
                a, b = (q_blk - e, q_blk - s) if rev else (s, e)
                hits = []
                outs, _ = E.SeedCalls([(a, b, rev)], 0, 2, hits_out=hits)
                want = [g for g in c["calls"] if g["interval"] == kk and g["rev"] == int(rev)]
                assert np.array_equal(outs[0], np.concatenate([g["hsps"] for g in want])), (G.case_id(c), "grouped", kk, rev)
                assert hits[0] == sum(g["num_hits"] for g in want)
    finally:
        E.set_max_hits(0)
        E.ShutdownProcessor()
        E.reset_option(None)
