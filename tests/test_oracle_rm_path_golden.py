"""CPU: the oracle's repeat-masker path -- table, host chunk loop, the fork's SeedAndFilter (u64 scans, plan over MAX_HITS, window flag,
reverse-strand flip, sort / unique / diagonal sort / diagonal unique / final sort), uint8 coverage, run extraction, .intervals text -- against
what the repeat masker binary's OWN FILES return and write when they run end to end (tests/golden/rm_path_golden.json; generator
tests/golden/make_rm_path_golden.py: the files compiled as they lie, CUDA runtime / thrust / TBB stood in for, kernels under SIMT emulation).
A second route, not a pin."""
import numpy as np
import pytest

import rm_path_golden as G
from host_model import rm_chunk_calls, rm_interval_lines
from rm_golden import SEG

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_the_oracles_rm_path_returns_and_writes_what_the_reference_files_do(oracle, c):
    O = oracle
    k = O.generate_shape_pos(c["shape"])
    seq = c["seq"].encode("ascii")
    L, bs, bl, span = len(seq), c["block_start"], c["block_len"], len(c["shape"])
    rc = O.rev_comp_ascii(seq, 0, L)                                     # repeat_masker_src/main.cpp:311
    rc_block_start = L - 1 - bs - (bl - 1)                               # seeder.cpp:49
    ref = O.encode(seq[bs:bs + bl])                                      # compress_string
    ref_rc = O.rev_comp_codes(ref)                                       # rev_comp_string (rm seed_filter.cu:120-168)
    index, pos = O.generate_seed_pos_table(seq, bs, bl, c["step"], span, k)
    assert O.max_hits_for_mem(c["total_global_mem"]) == c["max_hits"]
    names, starts, _ = c["chr"]
    split = 0
    for ti, t in enumerate(c["tasks"]):
        s, e, ws, we = t["interval"]
        calls = iter(t["calls"])
        allh = []
        for (rev, s0, s1) in rm_chunk_calls(s, e, bl, c["chunk"], c["strand"]):
            seeds = O.make_seeds(rc, rc_block_start, s0, s1, span, k, bool(c["transition"])) if rev else O.make_seeds(seq, bs, s0, s1, span, k, bool(c["transition"]))
            if seeds.size == 0:
                continue                                                 # seeder.cpp:101 / :139
            g = next(calls)
            assert (g["rev"], g["ref_start"], g["ref_end"], g["n_seeds"]) == (int(rev), ws, we, seeds.size)
            segs, st = O.seed_and_filter(ref, ref_rc if rev else ref, index, pos, seeds, c["sub_mat"], seed_size=span, xdrop=c["xdrop"], hspthresh=c["hspthresh"],
                                         noentropy=bool(c["noentropy"]), max_hits=c["max_hits"], rm=(bool(rev), ws, we))
            assert G.header(segs) == (g["num_hits"], g["n_hsps"]), (G.case_id(c), ti, rev, s0, s1)
            assert np.array_equal(segs[1:], g["hsps"]), (G.case_id(c), ti, rev, s0, s1)
            allh.append(segs[1:])
            split += g["num_hits"] >= c["max_hits"]
        assert next(calls, None) is None
        allh = np.concatenate(allh) if allh else np.zeros(0, dtype=SEG)
        runs = O.rm_coverage_intervals(allh, bl, c["M"])                 # seeder.cpp:153-188
        assert [[int(r["query_start"]), int(r["len"])] for r in runs] == t["runs"], (G.case_id(c), ti)
        text = "".join(rm_interval_lines(names, starts, bs, t["runs"], bool(c["markend"]))) if t["runs"] else None
        assert text == t["file"], (G.case_id(c), ti)                     # segment_printer.cpp:8-65
    if c["max_hits"] < 1 << 20 and c["strand"] == 3:
        assert split > 0


def test_the_golden_set_covers_the_rm_paths_corners():
    assert {c["strand"] for c in CASES} >= {2, 3} and {c["transition"] for c in CASES} == {0, 1} and {c["M"] for c in CASES} == {1, 2}
    assert any(c["block_start"] > 0 for c in CASES) and any(c["markend"] for c in CASES) and any(c["noentropy"] for c in CASES)
    assert sum(len(t["runs"]) for c in CASES for t in c["tasks"]) > 15
    for c in CASES:   # HSPs from minus-strand calls wherever that strand is walked, and at least one that is not the trivial diagonal
        mh = np.concatenate([k["hsps"] for t in c["tasks"] for k in t["calls"] if k["rev"]] or [np.zeros(0, dtype=SEG)])
        assert mh.size > 0, G.case_id(c)
        ph = np.concatenate([k["hsps"] for t in c["tasks"] for k in t["calls"] if not k["rev"]] or [np.zeros(0, dtype=SEG)])
        if c["strand"] & 1:
            assert np.count_nonzero(ph["ref_start"] != ph["query_start"]) > 0, G.case_id(c)
