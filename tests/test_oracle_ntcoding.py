"""CPU-only: the oracle's k-mer / shape / RevComp restatement is PINNED against the reference:
  1. the committed golden vectors generated from the real common/ntcoding.cpp (tests/golden/make_ntcoding_golden.py)
  2. the live reference object oracle/_ref/libntcoding_ref.so when it exists (authoring container or shipped .so)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "ntcoding_golden.json")) as f:
        return json.load(f)


def test_kmers_match_reference_golden(oracle, golden):
    n = 0
    for case in golden["kmer_cases"]:
        shape = case["shape"]
        assert oracle.generate_shape_pos(shape) == case["kmer_size"]
        assert [oracle.is_transition_at_pos(t) for t in range(case["kmer_size"])] == case["transition"]
        for s in case["seqs"]:
            got = [oracle.kmer_index_at_pos(s["seq"].encode(), p, len(shape)) for p in range(len(s["kmers"]))]
            assert got == s["kmers"]
            n += len(got)
    assert n > 1000


def test_revcomp_matches_reference_golden(oracle, golden):
    for c in golden["revcomp_cases"]:
        assert oracle.rev_comp_ascii(c["seq"].encode(), c["start"], c["len"]).decode() == c["rc"]


def test_invalid_kmer_rules(oracle):
    """Anything but upper-case ACGT invalidates the window (ntcoding.cpp:10-19,49-51); first care position is the
    most significant 2 bits (:56-58)."""
    oracle.generate_shape_pos("TTT0T00TT00T0T0TTTT")
    base = b"ACGTACGTACGTACGTACGTAC"
    assert oracle.kmer_index_at_pos(base, 0, 19) < (1 << 24)
    for bad in b"acgtNn&X-":
        for off in (0, 5, 18):  # also at don't-care positions
            s = bytearray(base)
            s[off] = bad
            assert oracle.kmer_index_at_pos(bytes(s), 0, 19) == 0x80000000
        s = bytearray(base)
        s[19] = bad  # just outside the span
        assert oracle.kmer_index_at_pos(bytes(s), 0, 19) != 0x80000000
    oracle.generate_shape_pos("T0T")
    assert oracle.kmer_index_at_pos(b"GAT", 0, 3) == (2 << 2) | 3


def test_live_reference_object_when_present(oracle):
    R = oracle.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built here (no /root/reference); golden vectors cover it")
    rng = np.random.default_rng(7)
    for shape in ("TTT0T00TT00T0T0TTTT", "TTT0T0TT00TT00T0T0TTTT", "1T0T1"):
        assert R.ref_GenerateShapePos(shape.encode()) == oracle.generate_shape_pos(shape)
        seq = bytes(rng.choice(np.frombuffer(b"ACGTACGTACGTACGTacgtN&", dtype=np.uint8), size=3000))
        for p in range(0, len(seq) - len(shape)):
            assert R.ref_GetKmerIndexAtPos(seq, p, len(shape)) == oracle.kmer_index_at_pos(seq, p, len(shape))
    seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn&", dtype=np.uint8), size=500))
    dst = C.create_string_buffer(400)
    R.ref_RevComp(dst, seq, 0, 50, 400)
    assert dst.raw == oracle.rev_comp_ascii(seq, 50, 400)
