"""GPU: the engine's repeat-masker entries against what the repeat masker binary's OWN FILES return and write when they run end to end
(tests/golden/rm_path_golden.json, generator tests/golden/make_rm_path_golden.py): sa_rm_seed_and_filter on the host loop's seed vectors
call by call (64-bit header + HSPs in the reference's order, MAX_HITS set to what the reference's arithmetic gives the generator's small
"GPU"), sa_rm_coverage_intervals on the collected HSPs, and sa_rm_mask_interval -- the whole interval task on the device -- against the
runs the reference's seeder returns.  A second route, not a pin."""
import numpy as np
import pytest

import rm_path_golden as G
from host_model import rm_chunk_calls
from rm_golden import SEG

pytestmark = pytest.mark.gpu

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_engine_returns_what_the_rm_reference_files_return(oracle, engine, c):
    E, O = engine, oracle
    seq = c["seq"].encode("ascii")
    arr = np.frombuffer(seq, dtype=np.uint8)
    L, bs, bl, span = len(seq), c["block_start"], c["block_len"], len(c["shape"])
    rc = O.rev_comp_ascii(seq, 0, L)
    rc_block_start = L - 1 - bs - (bl - 1)
    try:
        E.reset_option(None)
        E.InitializeInterface(1)
        k = E.GenerateShapePos(c["shape"])
        assert O.generate_shape_pos(c["shape"]) == k
        E.InitializeProcessor(bool(c["transition"]), c["chunk"], span, c["sub_mat"], c["xdrop"], c["hspthresh"], bool(c["noentropy"]))
        E.set_max_hits(c["max_hits"])
        keep = E.SendRefWriteRequest(arr, bs, bl)
        E.RmSendQueryWriteRequest()
        E.GenerateSeedPosTable(keep, bs, bl, c["step"], span, k)
        for ti, t in enumerate(c["tasks"]):
            s, e, ws, we = t["interval"]
            calls = iter(t["calls"])
            allh = []
            for (rev, s0, s1) in rm_chunk_calls(s, e, bl, c["chunk"], c["strand"]):
                seeds = O.make_seeds(rc, rc_block_start, s0, s1, span, k, bool(c["transition"])) if rev else O.make_seeds(seq, bs, s0, s1, span, k, bool(c["transition"]))
                if seeds.size == 0:
                    continue
                g = next(calls)
                got = E.RmSeedAndFilter(seeds, rev, ws, we)
                assert G.header(got) == (g["num_hits"], g["n_hsps"]), (G.case_id(c), ti, rev, s0, s1)
                assert np.array_equal(got[1:], g["hsps"]), (G.case_id(c), ti, rev, s0, s1)
                allh.append(got[1:])
            assert next(calls, None) is None
            allh = np.concatenate(allh) if allh else np.zeros(0, dtype=SEG)
            runs = E.RmCoverageIntervals(allh, bl, c["M"])
            assert [[int(a), int(b)] for a, b in zip(runs["query_start"], runs["len"])] == t["runs"], (G.case_id(c), ti)
            runs, tot = E.RmMaskInterval(s, e, ws, we, c["strand"], c["M"])   # the same task with the chunk loop on the device
            assert [[int(a), int(b)] for a, b in zip(runs["query_start"], runs["len"])] == t["runs"], (G.case_id(c), ti)
            assert (tot["num_hits"], tot["num_hsps"]) == (sum(k["num_hits"] for k in t["calls"]), sum(k["n_hsps"] for k in t["calls"]))
    finally:
        E.set_max_hits(0)
        E.RmClearQuery()
        E.ShutdownProcessor()
        E.reset_option(None)
