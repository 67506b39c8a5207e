"""GPU: sa_rm_seed_and_filter on the inputs of tests/golden/rm_golden.json gives the vector the reference's own repeat-masker device
code (under SIMT emulation) + comparators lead to: header (rm :857-861) and every record of `final`; and every record the engine
returns is one compress_output wrote (`reduced`: window flag, skip and reverse-strand flip already applied by reference text)."""
import numpy as np
import pytest

import rm_golden as G
from helpers import Case

pytestmark = pytest.mark.gpu

CASES = list(G.cases())


@pytest.fixture
def clean(engine):
    yield engine
    engine.RmClearQuery()
    engine.ShutdownProcessor()
    engine.reset_option(None)


@pytest.mark.parametrize("mode", ["default", "general path"])
def test_engine_equals_the_emulated_reference_on_the_golden_inputs(oracle, clean, mode):
    E, O = clean, oracle
    done = 0
    for seed in sorted({c["seed"] for c in CASES}):
        group = [c for c in CASES if c["seed"] == seed]
        for (hspthresh, noentropy) in sorted({(c["hspthresh"], c["noentropy"]) for c in group}):
            E.reset_option(None)
            if mode == "general path":
                E.set_option("no_td", 1)
            t = group[0]["target"]
            case = Case(t, t, chunk=250000, hspthresh=hspthresh, noentropy=bool(noentropy), sub_mat=group[0]["sub_mat"]).oracle_setup(O).engine_setup(E)
            E.RmSendQueryWriteRequest()
            rc_ascii = O.rev_comp_ascii(t.tobytes(), 0, t.size)
            for c in group:
                if (c["hspthresh"], c["noentropy"]) != (hspthresh, noentropy):
                    continue
                buf = rc_ascii if c["rev"] else t.tobytes()
                seeds = O.make_seeds(buf, 0, c["start"], c["end"], 19, case.kmer_size, True)
                got = E.RmSeedAndFilter(seeds, c["rev"], c["win_start"], c["win_end"])
                n_hits = int(got[0]["ref_start"]) | (int(got[0]["query_start"]) << 32)
                assert n_hits == c["hits"].size and int(got[0]["len"]) == c["final"].size
                body = got[1:]
                for f in ("ref_start", "query_start", "len", "score"):
                    assert np.array_equal(body[f], c["final"][f]), (G.case_id(c), f)
                red = {tuple(int(x) for x in r) for r in c["reduced"].tolist()}
                assert all(tuple(int(x) for x in r) in red for r in body.tolist())
                done += 1
            E.RmClearQuery()
            E.ShutdownProcessor()
    assert done == len(CASES)


def test_src_engine_equals_the_emulated_reference_on_the_golden_inputs(oracle, engine):
    """the src/ binary's golden set (tests/golden/src_golden.json): the drop-in entry on the host's seed words and the device-seeded
    entry both return the emulated reference's vector -- header {anchors, num_hits} and every record of `final`"""
    import src_golden as S
    E, O = engine, oracle
    cases = list(S.cases())
    done = 0
    try:
        for seed in sorted({c["seed"] for c in cases}):
            for key in sorted({(c["hspthresh"], c["noentropy"], c["transition"]) for c in cases if c["seed"] == seed}):
                group = [c for c in cases if c["seed"] == seed and (c["hspthresh"], c["noentropy"], c["transition"]) == key]
                t, q = group[0]["target"], group[0]["query"]
                E.reset_option(None)
                case = Case(t, q, chunk=250000, hspthresh=key[0], noentropy=bool(key[1]), transition=bool(key[2]), sub_mat=group[0]["sub_mat"]).oracle_setup(O).engine_setup(E)
                assert np.array_equal(E.copy_ref_codes(), group[0]["t_codes"])
                assert np.array_equal(E.copy_query_codes(0, False), group[0]["q_codes"]) and np.array_equal(E.copy_query_codes(0, True), group[0]["q_rc_codes"])
                for c in group:
                    seeds = case.host_seeds(c["start"], c["end"], bool(c["rev"]))
                    for got in (E.SeedAndFilter(seeds, bool(c["rev"]), 0), E.SeedAndFilterRange(c["start"], c["end"], bool(c["rev"]), 0)):
                        assert int(got[0]["len"]) == c["final"].size and int(got[0]["score"]) == c["hits"].size
                        for f in ("ref_start", "query_start", "len", "score"):
                            assert np.array_equal(got[1:][f], c["final"][f]), (S.case_id(c), f)
                    done += 1
                E.ShutdownProcessor()
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)
    assert done == len(cases)


def test_device_coverage_runs_equal_the_reference_seeders(engine):
    """sa_rm_coverage_intervals (coverage.hip: difference array + scans, depth mod 256) on the HSPs the reference's own seeder was handed
    (tests/golden/rm_host_golden.json: repeat_masker_src/seeder.cpp compiled as it lies) gives the runs that seeder returned -- the uint8
    wrap under piles of 300 HSPs and the dropped open run at the block end included"""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rm_host_golden.json")))["cases"]
    engine.InitializeInterface(1)
    engine.GenerateShapePos(G.SHAPE)
    engine.InitializeProcessor(True, 250000, 19, np.array(cases[0].get("sub_mat") or list(CASES[0]["sub_mat"]), dtype=np.int32), 910, 3000, False)
    n = 0
    for c in cases:
        for t in c["tasks"]:
            hs = [G._rows(g["hsps"], G.SEG) for g in t["calls"]]
            allh = np.concatenate(hs) if hs else np.zeros(0, dtype=G.SEG)
            got = engine.RmCoverageIntervals(allh, c["block_len"], c["M"])
            assert [[int(r["query_start"]), int(r["len"])] for r in got] == t["runs"]
            n += len(t["runs"])
    engine.ShutdownProcessor()
    assert n > 30
