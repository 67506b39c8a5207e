"""GPU: randomized parity -- random substitution matrices (asymmetric, positive off-diagonals, huge penalties),
random xdrop / hspthresh / entropy switch, sequences over all 8 codes, low-complexity repeats with giant buckets."""
import numpy as np
import pytest

from helpers import Case, canonical_pos_table, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def random_matrix(rng):
    m = rng.integers(-150, 30, size=(8, 8)).astype(np.int32)
    for i in range(4):
        m[i, i] = int(rng.integers(50, 130))
    if rng.random() < 0.5:  # transitions mildly negative like HOXD70
        for a, b in ((0, 2), (2, 0), (1, 3), (3, 1)):
            m[a, b] = int(rng.integers(-40, 20))
    m[4:, :] = rng.integers(-1200, -50, size=(4, 8))
    m[:, 4:] = rng.integers(-1200, -50, size=(8, 4))
    if rng.random() < 0.3:
        m[6, 6] = int(rng.integers(-10, 60))  # X-X reward (iupac-style)
    m[7, :] = m[:, 7] = -int(rng.integers(2000, 20000))
    return m.reshape(64)


@pytest.mark.parametrize("seed", range(10))
def test_random_scoring_systems(oracle, engine, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(20000, 50000))
    t, q = synth.make_pair(n, 2000 + seed, 3000 + seed, sub_rate=float(rng.uniform(0.03, 0.2)),
                           mask_frac=float(rng.uniform(0, 0.15)), records=int(rng.integers(1, 4)),
                           indel_every=int(rng.integers(100, 600)), n_runs=int(rng.integers(0, 3)),
                           invert_block=3000)
    # IUPAC / junk characters -> code X.  Target only: in the QUERY they trigger hazard H14 (separate test below)
    t[rng.integers(0, t.size, size=20)] = rng.choice(np.frombuffer(b"RYKMSW-*", dtype=np.uint8), size=20)
    xdrop = int(rng.choice([0, 50, 300, 910, 2500]))
    hspthresh = int(rng.choice([600, 1500, 3000, 5000]))
    c = Case(t, q, xdrop=xdrop, hspthresh=hspthresh, noentropy=bool(rng.random() < 0.4), chunk=int(rng.integers(8000, 30000)),
             sub_mat=random_matrix(rng), transition=bool(rng.random() < 0.7))
    c.oracle_setup(oracle).engine_setup(engine)
    try:
        total = 0
        for rev in (False, True):
            for (s, e) in c.chunks():
                seeds = c.host_seeds(s, e, rev)
                if seeds.size == 0:
                    continue
                want, _ = c.oracle_saf(seeds, rev)
                assert seg_equal(c.E.SeedAndFilter(seeds, rev, 0), want), (seed, rev, s, e)
                assert seg_equal(c.E.SeedAndFilterRange(s, e, rev, 0), want)
                total += want.size - 1
    finally:
        engine.ShutdownProcessor()


def test_low_complexity_giant_buckets(oracle, engine):
    """Poly-A and (CA)n tracts: buckets with thousands of positions (left in arrival order by the table build) and seeds
    with thousands of hits each; output must still be bit-identical."""
    t = synth.random_dna(60000, 5)
    q = synth.random_dna(8000, 6)
    t[10000:13000] = ord("A")
    t[30000:32000] = np.tile(np.frombuffer(b"CA", dtype=np.uint8), 1000)
    q[1000:1400] = ord("A")
    q[5000:5300] = np.tile(np.frombuffer(b"CA", dtype=np.uint8), 150)
    c = Case(t, q, chunk=4000).oracle_setup(oracle).engine_setup(engine)
    try:
        idx = c.E.copy_index_table()
        assert np.array_equal(idx, c.o_index)
        sizes = np.diff(np.concatenate([[0], idx.astype(np.int64)]))
        assert sizes.max() > 1000
        assert np.array_equal(canonical_pos_table(idx, c.E.copy_pos_table()), c.o_pos)
        total = 0
        for rev in (False, True):
            for (s, e) in c.chunks():
                seeds = c.host_seeds(s, e, rev)
                want, st = c.oracle_saf(seeds, rev)
                assert seg_equal(c.E.SeedAndFilter(seeds, rev, 0), want)
                total += st["num_hits"]
        assert total > 500000
    finally:
        engine.ShutdownProcessor()


def test_h14_junk_characters_in_the_query(oracle, engine):
    """Hazard H14 (found by this suite): the reference's host RevComp (common/ntcoding.cpp:63-105) does not advance its
    write index for characters outside {ACGTacgtNn&}, so the host's minus-strand arena is SHIFTED after the first such
    character while the device mirrors the block correctly (src/seed_filter.cu:153).  The drop-in entry receives the
    host's (shifted) seed words and must reproduce that exactly; the additive device seeder sees the mirrored block."""
    t, q = synth.make_pair(40000, 71, 72, sub_rate=0.08, invert_frac=0.5, invert_block=4000)
    q[[5000, 21000]] = ord("R")
    c = Case(t, q, chunk=40000).oracle_setup(oracle).engine_setup(engine)
    try:
        assert bytes(c.query_rc_ascii[-2:]) == b"\0\0"  # two characters dropped by the reference RevComp: arena is 2 short
        (s, e) = c.chunks()[0]
        seeds = c.host_seeds(s, e, True)
        want, _ = c.oracle_saf(seeds, True)
        assert seg_equal(c.E.SeedAndFilter(seeds, True, 0), want)  # bit-identical to the reference behaviour
        # device seeding == host seeding on the correctly mirrored block
        mirrored = synth.reverse_complement(q)
        mirrored[mirrored == 0] = ord("R")  # the translate table leaves unknown characters as 0
        seeds_ok = oracle.make_seeds(mirrored.tobytes(), 0, s, e, 19, c.kmer_size, True)
        want_ok, _ = c.oracle_saf(seeds_ok, True)
        assert np.array_equal(c.E.device_make_seeds(s, e, True, 0), seeds_ok)
        assert seg_equal(c.E.SeedAndFilterRange(s, e, True, 0), want_ok)
        assert want_ok.size > 1
    finally:
        engine.ShutdownProcessor()
