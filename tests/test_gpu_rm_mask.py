"""GPU: repeat-masker post-processing on the device against the oracle.
  * sa_rm_coverage_intervals == repeat_masker_src/seeder.cpp:153-188 on arbitrary HSP sets (uint8 wrap, M, tiles);
  * sa_rm_mask_interval == the whole seeder_body::operator() (seeder.cpp:28-195): chunk loop, both strands (the
    minus-strand chunk derived from the plus-strand chunk end), window filter, coverage, runs."""
import numpy as np
import pytest

from helpers import Case
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def as_list(iv):
    return [(int(a), int(b)) for a, b in zip(iv["query_start"], iv["len"])]


def random_hsps(O, rng, n, block_len, max_len):
    h = np.zeros(n, dtype=O.SEG_DTYPE)
    h["len"] = rng.integers(0, max_len, n)
    h["query_start"] = rng.integers(0, block_len - max_len, n)
    return h


@pytest.fixture(scope="module")
def rm_case(oracle, engine):
    unit = synth.random_dna(600, 77)
    t = synth.random_dna(200000, 15)
    rng = np.random.default_rng(3)
    for i in range(90):  # a diverged repeat family, both orientations
        p = int(rng.integers(0, t.size - 700))
        cp = synth.mutate(unit, 500 + i, 0.05)
        t[p:p + cp.size] = cp if i % 3 else synth.reverse_complement(cp)
    t = synth.soft_mask(t, 5, 0.04)
    c = Case(t, t, chunk=30000).oracle_setup(oracle).engine_setup(engine)
    engine.RmSendQueryWriteRequest()
    yield c
    engine.RmClearQuery()
    engine.ShutdownProcessor()


@pytest.mark.parametrize("M", [0, 1, 2, 5, 255, 256])
def test_coverage_intervals_random_sets(oracle, rm_case, M):
    E, O = rm_case.E, oracle
    rng = np.random.default_rng(40 + M)
    for n in (0, 1, 7, 3000):
        h = random_hsps(O, rng, n, 200000, 900)
        assert as_list(E.RmCoverageIntervals(h, 200000, M)) == as_list(O.rm_coverage_intervals(h, 200000, M)), (n, M)


def test_coverage_uint8_wrap(oracle, rm_case):
    E, O = rm_case.E, oracle
    h = np.zeros(300, dtype=O.SEG_DTYPE)
    h["query_start"] = 1000
    h["len"] = 77
    h["query_start"][256:] = 1010  # depth 256 (reads 0) on 1000..1009, 300 (reads 44) on 1010..1076, 44 on 1077..1086
    for M in (1, 44, 45, 200):
        assert as_list(E.RmCoverageIntervals(h, 5000, M)) == as_list(O.rm_coverage_intervals(h, 5000, M)), M
    assert as_list(E.RmCoverageIntervals(h, 5000, 1)) == [(1010, 77)]


def test_coverage_is_clean_between_calls_and_blocks(oracle, rm_case):
    """the difference array is re-zeroed over the touched range only; a later, larger block re-allocates it"""
    E, O = rm_case.E, oracle
    rng = np.random.default_rng(9)
    a = random_hsps(O, rng, 500, 50000, 300)
    b = random_hsps(O, rng, 20, 50000, 300)
    assert as_list(E.RmCoverageIntervals(a, 50000, 1)) == as_list(O.rm_coverage_intervals(a, 50000, 1))
    assert as_list(E.RmCoverageIntervals(b, 50000, 1)) == as_list(O.rm_coverage_intervals(b, 50000, 1))
    c = random_hsps(O, rng, 500, 900000, 300)
    assert as_list(E.RmCoverageIntervals(c, 900000, 2)) == as_list(O.rm_coverage_intervals(c, 900000, 2))


def test_coverage_across_scan_tiles(oracle, rm_case):
    """a touched range wider than one 2^26-position scan tile: depth and run indices are carried between tiles"""
    E, O = rm_case.E, oracle
    L = (1 << 26) * 2 + 12345
    rng = np.random.default_rng(5)
    h = np.zeros(4000, dtype=O.SEG_DTYPE)
    h["query_start"] = rng.integers(0, L - 5000, h.size)
    h["len"] = rng.integers(1, 4000, h.size)
    edge = 1 << 26
    h["query_start"][:4] = [edge - 10, edge - 1, edge, 2 * edge - 3]  # runs that straddle / touch the tile edges
    h["len"][:4] = [20, 1, 5, 3000]
    h["query_start"][4], h["len"][4] = 17, 2 * edge + 100  # one HSP over both edges (depth carried, M=2 runs inside)
    for M in (1, 2):
        assert as_list(E.RmCoverageIntervals(h, L, M)) == as_list(O.rm_coverage_intervals(h, L, M)), M


def model_mask_interval(c, O, start_pos, end_pos, ref_start, ref_end, strands, M):
    """seeder_body::operator() of the repeat masker restated with the oracle's pieces (seeder.cpp:73-188)."""
    t = c.target
    L = t.size
    rc_ascii = np.frombuffer(O.rev_comp_ascii(t.tobytes(), 0, L), dtype=np.uint8)
    o_rc = O.rev_comp_codes(c.o_ref)
    end_pos_rc = L - 1 - start_pos
    hsps = []
    tot = dict(num_seeds=0, num_hits=0, num_hsps=0)
    for i in range(start_pos, end_pos, c.chunk):
        start, end = i, min(i + c.chunk, end_pos)
        for rev in (False, True):
            if not (strands & (2 if rev else 1)):
                continue
            s0, s1 = start, end
            if rev:  # :118-119
                s0 = L - 1 - end
                s1 = min(s0 + c.chunk, end_pos_rc)
            s1 = min(s1, L - c.seed_size + 1)
            buf = rc_ascii if rev else t
            seeds = O.make_seeds(buf.tobytes(), 0, s0, s1, c.seed_size, c.kmer_size, c.transition)
            if seeds.size == 0:
                continue
            segs, st = O.seed_and_filter(c.o_ref, o_rc if rev else c.o_ref, c.o_index, c.o_pos, seeds, c.sub_mat,
                                         seed_size=c.seed_size, xdrop=c.xdrop, hspthresh=c.hspthresh, noentropy=c.noentropy,
                                         rm=(rev, ref_start, ref_end))
            tot["num_seeds"] += int(seeds.size)
            tot["num_hits"] += int(st["num_hits"])
            tot["num_hsps"] += int(segs.size - 1)
            hsps.append(segs[1:])
    allh = np.concatenate(hsps) if hsps else np.zeros(0, dtype=O.SEG_DTYPE)
    return O.rm_coverage_intervals(allh, L, M), tot


@pytest.mark.parametrize("strands", [1, 2, 3])
def test_mask_interval_matches_reference_loop(oracle, rm_case, strands):
    c, E, O = rm_case, rm_case.E, oracle
    L = c.target.size
    for (s, e, ws, we, M) in ((0, L - 19, 0, L, 1), (40000, 130000, 20000, 150000, 1), (0, L - 19, 0, L, 2),
                              (100000, 100001, 0, L, 1), (50000, 50000, 0, L, 1),
                              (100000, 160000, 0, 90000, 1), (0, 70000, 120000, L, 1), (100000, 160000, 0, 90000, 2)):
        want, wt = model_mask_interval(c, O, s, e, ws, we, strands, M)
        got, gt = E.RmMaskInterval(s, e, ws, we, strands, M)
        assert as_list(got) == as_list(want), (s, e, ws, we, M, as_list(got)[:4], as_list(want)[:4])
        assert gt == wt
    # a window that excludes the interval itself leaves only the planted family (no trivial self-alignment)
    want, _ = model_mask_interval(c, O, 100000, 160000, 0, 90000, 3, 1)
    assert want.size > 10
