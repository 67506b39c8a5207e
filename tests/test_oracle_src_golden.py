"""CPU: the oracle's src/ path, stage by stage, against the lists the reference's own device code leaves behind (common/
seed_filter_interface.cu:18-47 and src/seed_filter.cu:47-680 executed under SIMT emulation: tests/golden/make_src_golden.py).

  t_codes / q_codes / q_rc_codes   compress_string, compress_string_rev_comp   == orc_encode / orc_encode_rev_comp (a-3)
  hits      find_num_hits + find_hits (:157-230)   == the oracle's hit list: slot order inside a seed word's bucket, seed_size offsets (a-6)
  ext       find_hsps (:232-652)                   == records + done flags (a-8, a second input family next to find_hsps_golden.json)
  reduced   compress_output (:654-680)             == order-preserving compaction (a-9)
  final     hspComp / hspEqual / hspCompLastz (:47-108) under std::stable_sort + adjacent-pair unique == the oracle's output vector and its
            header {len = anchors, score = num_hits} (:806-809)

A second route, not a pin: the emulation stands in for the CUDA runtime."""
import numpy as np
import pytest

import src_golden as G

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_oracle_stages_equal_the_emulated_reference_kernels(oracle, c):
    O = oracle
    k = O.generate_shape_pos(G.SHAPE)
    t, q = c["target"], c["query"]
    tc = O.encode(t.tobytes())
    qc, qrc = O.encode_rev_comp(q.tobytes())
    assert np.array_equal(tc, c["t_codes"]) and np.array_equal(qc, c["q_codes"]) and np.array_equal(qrc, c["q_rc_codes"])
    index, pos = O.generate_seed_pos_table(t.tobytes(), 0, t.size, 1, 19, k)
    buf = O.rev_comp_ascii(q.tobytes(), 0, q.size) if c["rev"] else q.tobytes()
    seeds = O.make_seeds(buf, 0, c["start"], c["end"], 19, k, bool(c["transition"]))
    assert seeds.size == c["num_seeds"]
    segs, tr = O.seed_and_filter_traced(tc, qrc if c["rev"] else qc, index, pos, seeds, c["sub_mat"], xdrop=c["xdrop"], hspthresh=c["hspthresh"],
                                        noentropy=bool(c["noentropy"]))
    for f in ("ref_start", "query_start", "len", "score"):
        assert np.array_equal(tr["hits"][f], c["hits"][f]), ("hits", f)
        assert np.array_equal(tr["ext"][f], c["ext"][f]), ("ext", f)
        assert np.array_equal(tr["reduced"][f], c["reduced"][f]), ("reduced", f)
        assert np.array_equal(segs[1:][f], c["final"][f]), ("final", f)
    assert np.array_equal(tr["done"].astype(np.uint32), c["ext"]["done"])
    assert int(segs[0]["len"]) == c["final"].size and int(segs[0]["score"]) == c["hits"].size   # :806-809


def test_the_golden_set_has_both_strands_and_real_ties():
    assert {c["rev"] for c in CASES} == {0, 1} and {c["transition"] for c in CASES} == {0, 1}
    assert sum(c["hits"].size for c in CASES) > 3000 and sum(c["reduced"].size - c["final"].size for c in CASES) > 1000
    assert all(c["final"].size > 0 for c in CASES)
