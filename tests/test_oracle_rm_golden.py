"""CPU: the oracle's repeat-masker variant, stage by stage, against the lists the reference's own device code leaves behind
(repeat_masker_src/seed_filter.cu:45-722 executed under SIMT emulation: tests/golden/make_rm_golden.py).

  rc_codes  rev_comp_string (:137-167)            == orc_rev_comp_codes
  hits      find_num_hits + find_hits (:169-248)  == the oracle's hit list: slot order, seed_size offsets, the window flag of :239-244
  ext       find_hsps (:250-689)                  == records + done flags, flagged hits skipped (:305-333: total 0, never an anchor)
  reduced   compress_output (:691-722)            == order-preserving compaction, reverse-strand flip of :705-708
  final     the reference's comparators (:45-135) under std::stable_sort + adjacent-pair unique (the harness's reading of thrust)
            == the oracle's output vector behind its header (rm :857-861)

A second route, not a pin: the emulation stands in for the CUDA runtime."""
import numpy as np
import pytest

import rm_golden as G

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_oracle_stages_equal_the_emulated_reference_kernels(oracle, c):
    O = oracle
    k = O.generate_shape_pos(G.SHAPE)
    t = c["target"]
    L = t.size
    codes = O.encode(t.tobytes())
    rc = O.rev_comp_codes(codes)
    assert np.array_equal(rc, c["rc_codes"])
    index, pos = O.generate_seed_pos_table(t.tobytes(), 0, L, 1, 19, k)
    buf = O.rev_comp_ascii(t.tobytes(), 0, L) if c["rev"] else t.tobytes()
    seeds = O.make_seeds(buf, 0, c["start"], c["end"], 19, k, True)
    assert seeds.size == c["num_seeds"]
    segs, tr = O.seed_and_filter_traced(codes, rc if c["rev"] else codes, index, pos, seeds, c["sub_mat"], xdrop=c["xdrop"],
                                        hspthresh=c["hspthresh"], noentropy=bool(c["noentropy"]), rm=(c["rev"], c["win_start"], c["win_end"]))
    for f in ("ref_start", "query_start", "len", "score"):
        assert np.array_equal(tr["hits"][f], c["hits"][f]), ("hits", f)
        assert np.array_equal(tr["ext"][f], c["ext"][f]), ("ext", f)
        assert np.array_equal(tr["reduced"][f], c["reduced"][f]), ("reduced", f)
        assert np.array_equal(segs[1:][f], c["final"][f]), ("final", f)
    assert np.array_equal(tr["done"].astype(np.uint32), c["ext"]["done"])
    # what the stages are there for
    flagged = c["hits"]["score"] < 0
    assert np.array_equal(flagged, ~((c["hits"]["ref_start"] >= c["win_start"]) & (c["hits"]["ref_start"] <= c["win_end"])))
    assert not np.any(c["ext"]["done"][flagged]) and np.all(c["ext"]["score"][flagged] == 0)
    n_hits = int(segs[0]["ref_start"]) | (int(segs[0]["query_start"]) << 32)   # rm :857-861
    assert n_hits == c["hits"].size and int(segs[0]["len"]) == c["final"].size


def test_the_golden_set_reaches_the_forks_three_differences():
    flagged = sum(int(np.count_nonzero(c["hits"]["score"] < 0)) for c in CASES)
    inside = sum(int(np.count_nonzero(c["hits"]["score"] == 0)) for c in CASES)
    rev_anchors = sum(c["reduced"].size for c in CASES if c["rev"])
    fwd_anchors = sum(c["reduced"].size for c in CASES if not c["rev"])
    dropped = sum(c["reduced"].size - c["final"].size for c in CASES)
    assert flagged > 300 and inside > 1000 and rev_anchors > 10 and fwd_anchors > 500 and dropped > 100
