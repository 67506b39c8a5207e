"""GPU: the HIP extension kernels (packed X-drop filter -> chain grouping -> exact segmented-scan kernel -> entropy kernel)
against the committed outputs of the REFERENCE's own find_hsps kernel text under SIMT emulation
(tests/golden/find_hsps_golden.json, see tests/test_oracle_find_hsps_golden.py), through sa_extend_hits."""
import numpy as np
import pytest

from test_oracle_find_hsps_golden import CASES

pytestmark = pytest.mark.gpu

ASCII = np.frombuffer(b"ACGTaNR&", dtype=np.uint8)  # codes A0 C1 G2 T3 L4 N5 X6 E7 (common/seed_filter_interface.cu:18-47)
SHAPE = "TTT0T00TT00T0T0TTTT"


def setup(E, c):
    E.InitializeInterface(1)
    E.GenerateShapePos(SHAPE)
    E.InitializeProcessor(True, 250000, 19, np.array(c["sub_mat"], dtype=np.int32), c["xdrop"], c["hspthresh"], bool(c["noentropy"]))
    t, q = ASCII[c["ref_codes"]], ASCII[c["query_codes"]]
    keep = E.SendRefWriteRequest(t, 0, t.size)
    E.SendQueryWriteRequest(q, 0, q.size, 0)
    assert np.array_equal(E.copy_ref_codes(), c["ref_codes"]) and np.array_equal(E.copy_query_codes(0, False), c["query_codes"])
    return keep


def as_set(recs):
    return {(int(a), int(b), int(l), int(s)) for a, b, l, s in recs}


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_bulk_extension_equals_reference_kernel_output(engine, idx):
    """All 1024 anchors of a case in one launch: the set of passing records equals the reference kernel's (the chain
    shortcut may merge exact duplicates, nothing else may differ)."""
    E, c = engine, CASES[idx]
    keep = setup(E, c)
    try:
        got = E.ExtendHits(c["hits_a"], False, 0)
        want = c["out_a"][c["out_a"][:, 4] == 1][:, :4]
        assert as_set(zip(got["ref_start"], got["query_start"], got["len"], got["score"])) == as_set(want)
        assert got.size <= want.shape[0]
    finally:
        del keep
        E.ShutdownProcessor()


@pytest.mark.parametrize("idx", [0, 1, 6, 7])
def test_single_anchor_extension_equals_reference_kernel_output(engine, idx):
    """One anchor per call: every hit's own verdict and record (entropy on: cases 0 and 6; off: 1 and 7; default and
    alternative xdrop / hspthresh)."""
    E, c = engine, CASES[idx]
    keep = setup(E, c)
    try:
        for (r, q), want in zip(c["hits_a"], c["out_a"]):
            got = E.ExtendHits(np.array([[r, q]], dtype=np.uint32), False, 0)
            if want[4] == 0:
                assert got.size == 0, (int(r), int(q), got)
            else:
                assert got.size == 1 and tuple(int(x) for x in got[0]) == tuple(int(x) for x in want[:4]), (int(r), int(q), got, want)
    finally:
        del keep
        E.ShutdownProcessor()
