"""CPU: the oracle's whole path -- alphabet, seed position table, host seeding loop, SeedAndFilter with its iteration plan, find_hsps, the
sort / unique / sort chain -- against what the reference's OWN FILES return when they run end to end (tests/golden/path_golden.json:
src/seed_filter.cu, common/seed_filter_interface.cu, common/seed_pos_table.cu, common/ntcoding.cpp and src/seeder.cpp compiled as they lie,
CUDA runtime / thrust / TBB stood in for, kernels under SIMT emulation; tests/golden/make_path_golden.py).  Every g_SeedAndFilter call of
the run: the header (HSPs, seed hits) and every HSP in the reference's order; MAX_HITS from the reference's own arithmetic on a 32-128 KiB
"GPU", so that most calls of cases 1-4 run in several iterations (src/seed_filter.cu:720-744); case 5 is a 60 kbp x 30 kbp pair with 5000-base chunks
(65 k seed words per call, six chunks per interval and strand).  A second route, not a pin."""
import numpy as np
import pytest

import path_golden as G
from segalign_amd import shard

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_the_oracles_path_returns_what_the_reference_files_return(oracle, c):
    O = oracle
    span, ts, tl, qs, ql = len(c["shape"]), c["t_start"], c["t_len"], c["q_start"], c["q_len"]
    k = O.generate_shape_pos(c["shape"])
    t_arena, q_arena = c["target_arena"].tobytes(), c["query_arena"].tobytes()
    ref_codes = O.encode(t_arena[ts:ts + tl])                                        # compress_string (seed_filter_interface.cu:18-47)
    index, pos = O.generate_seed_pos_table(t_arena, ts, tl, c["step"], span, k)      # seed_pos_table.cu:49-109
    fw_codes, rc_codes = O.encode_rev_comp(q_arena[qs:qs + ql])                      # compress_string_rev_comp (seed_filter.cu:111-156)
    rc_block = O.rev_comp_ascii(q_arena, qs, ql)                                     # src/main.cpp:377
    assert O.max_hits_for_mem(c["total_global_mem"]) == c["max_hits"]                # seed_filter.cu:833-841 on the generator's "GPU"
    calls = iter(c["calls"])
    split = 0
    for kk, rev, a, b in G.chunk_calls(c, shard):
        seeds = O.make_seeds(rc_block, 0, a, b, span, k, bool(c["transition"])) if rev else O.make_seeds(q_arena, qs, a, b, span, k, bool(c["transition"]))
        if seeds.size == 0:
            continue                                                                 # seeder.cpp:76 / :111
        g = next(calls)
        assert (g["interval"], g["rev"], g["n_seeds"]) == (kk, int(rev), seeds.size)
        segs, st = O.seed_and_filter(ref_codes, rc_codes if rev else fw_codes, index, pos, seeds, c["sub_mat"], span, c["xdrop"], c["hspthresh"],
                                     bool(c["noentropy"]), max_hits=c["max_hits"])
        assert (int(segs[0]["len"]), int(segs[0]["score"])) == (g["n_hsps"], g["num_hits"]), (G.case_id(c), kk, rev, a, b)
        assert segs.size - 1 == g["n_hsps"]
        assert np.array_equal(segs[1:], g["hsps"]), (G.case_id(c), kk, rev, a, b)
        split += g["num_hits"] >= c["max_hits"]
    assert next(calls, None) is None
    if c["max_hits"] <= 512:
        assert split > 0                                                             # calls that took the several-iteration plan


def test_the_golden_set_covers_the_paths_corners():
    assert sum(sum(k["n_hsps"] for k in c["calls"]) for c in CASES) > 300
    assert {c["strand"] for c in CASES} == {1, 2, 3} and {c["transition"] for c in CASES} == {0, 1}
    assert any(c["t_start"] and c["q_start"] for c in CASES) and any(c["step"] > 1 for c in CASES) and any(c["noentropy"] for c in CASES)
    assert {len(c["shape"]) for c in CASES} == {19, 22}
    for c in CASES:   # HSPs on both strands wherever both are walked
        for rev in (0, 1):
            if c["strand"] & (2 if rev else 1):
                assert sum(k["n_hsps"] for k in c["calls"] if k["rev"] == rev) > 0, (G.case_id(c), rev)
