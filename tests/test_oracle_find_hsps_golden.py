"""CPU: the oracle's X-drop extension against the committed outputs of the REFERENCE's own find_hsps kernel text
(src/seed_filter.cu:232-652) executed under a SIMT emulation -- tests/golden/find_hsps_golden.json, generator
tests/golden/make_find_hsps_golden.py.  10 240 hits: random, poly-A and CT-repeat homology
islands (many HSPs inside the entropy band), L/N/X/E codes, sequence corners, entropy on and off, two parameter sets."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "find_hsps_golden.json")


def load_cases():
    d = json.load(open(GOLDEN))
    for c in d["cases"]:
        c["ref_codes"] = np.frombuffer(c["ref"].encode(), dtype=np.uint8) - ord("0")
        c["query_codes"] = np.frombuffer(c["query"].encode(), dtype=np.uint8) - ord("0")
        c["hits_a"] = np.array(c["hits"], dtype=np.uint32)
        c["out_a"] = np.array(c["out"], dtype=np.int64)
    return d["cases"]


CASES = load_cases()


def test_vector_file_shape():
    assert len(CASES) == 10 and sum(c["hits_a"].shape[0] for c in CASES) >= 4000
    assert {c["noentropy"] for c in CASES} == {0, 1}
    band = sum(int(np.count_nonzero((c["out_a"][:, 4] == 1) & (c["out_a"][:, 3] <= 3 * c["hspthresh"]))) for c in CASES)
    assert band > 1000  # the entropy branch (:608-625) fires on a large share of the passing hits


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_oracle_extension_equals_reference_kernel_output(oracle, idx):
    c = CASES[idx]
    sub_mat = np.array(c["sub_mat"], dtype=np.int32)
    bad = 0
    for (r, q), want in zip(c["hits_a"], c["out_a"]):
        ok, rec, _ = oracle.extend_hit(c["ref_codes"], c["query_codes"], sub_mat, int(r), int(q), xdrop=c["xdrop"],
                                       hspthresh=c["hspthresh"], noentropy=bool(c["noentropy"]))
        got = (rec[0], rec[1], rec[2], rec[3], int(ok))
        if got != tuple(int(x) for x in want):
            bad += 1
            assert bad < 5, (idx, int(r), int(q), got, want.tolist())
    assert bad == 0
