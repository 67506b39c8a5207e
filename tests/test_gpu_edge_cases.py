"""GPU parity on the edge cases and parameter corners of the path (the reference has no tests; these follow the
hazards H1-H13 of SURVEY.md 8 and the parameter surface of src/main.cpp:61-124)."""
import threading

import numpy as np
import pytest

from helpers import Case, canonical_pos_table, seg_equal, SHAPE_12OF19
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def run_all_chunks(c, rev_list=(False, True), max_hits=None, check_range=True):
    n = 0
    for rev in rev_list:
        for (s, e) in c.chunks():
            seeds = c.host_seeds(s, e, rev)
            if seeds.size == 0:
                continue
            want, st = c.oracle_saf(seeds, rev, **({"max_hits": max_hits} if max_hits else {}))
            # the reference asserts num_seeds <= MAX_SEEDS = 13 * chunk (seed_filter.cu:688-692,836-839), which a
            # 14of22 seed with transitions (15 words per position) exceeds on a chunk without masked positions; the
            # drop-in entry keeps that behaviour, the device seeder has no such limit
            if seeds.size <= (13 if c.transition else 1) * c.chunk:
                got = c.E.SeedAndFilter(seeds, rev, 0)
                assert seg_equal(got, want), (rev, s, e, got[:3], want[:3])
            if check_range:
                assert seg_equal(c.E.SeedAndFilterRange(s, e, rev, 0), want)
            n += want.size - 1
    return n


@pytest.fixture
def engine_clean(engine):
    yield engine
    engine.ShutdownProcessor()
    engine.set_max_hits(0)


def test_empty_and_hitless_calls(oracle, engine_clean):
    t = synth.random_dna(50000, 1)
    q = synth.random_dna(30000, 2)  # unrelated: (almost) no hits in a 50 kb target
    c = Case(t, q, chunk=30000).oracle_setup(oracle).engine_setup(engine_clean)
    out = c.E.SeedAndFilter(np.zeros(0, dtype=np.uint64), False, 0)
    assert out.size == 1 and tuple(out[0]) == (0, 0, 0, 0)  # header only (seed_filter.cu:806-809)
    run_all_chunks(c)
    # a seed vector whose keys hit nothing at all
    seeds = c.host_seeds(0, 2000, False)
    lonely = np.array([s for s in seeds if (c.o_index[int(s >> np.uint64(32))] - (c.o_index[int(s >> np.uint64(32)) - 1] if int(s >> np.uint64(32)) else 0)) == 0][:50], dtype=np.uint64)
    want, st = c.oracle_saf(lonely, False)
    assert st["num_hits"] == 0 and seg_equal(c.E.SeedAndFilter(lonely, False, 0), want)


def test_single_hit_bearing_seed_and_all_hits_in_seed0(oracle, engine_clean):
    """Hazard H5: the reference indexes prefix[-1] here; the engine treats the empty iteration as 'no hits'."""
    t, q = synth.make_pair(60000, 3, 4, sub_rate=0.05)
    c = Case(t, q, chunk=60000).oracle_setup(oracle).engine_setup(engine_clean)
    seeds = c.host_seeds(0, q.size - 19, False)
    keys = (seeds >> np.uint64(32)).astype(np.int64)
    cnt = c.o_index[keys].astype(np.int64) - np.where(keys > 0, c.o_index[np.maximum(keys - 1, 0)], 0)
    with_hits = seeds[cnt > 0]
    without = seeds[cnt == 0]
    assert with_hits.size and without.size
    for vec in (with_hits[:1],                                   # all hits in seed 0
                np.concatenate([without[:5], with_hits[:1]]),    # one hit-bearing seed at the end
                np.concatenate([with_hits[:1], without[:5]]),    # ... at the start, followed by empty seeds
                np.concatenate([without[:3], with_hits[:1], without[3:6], with_hits[1:2]])):
        want, _ = c.oracle_saf(vec, False)
        assert seg_equal(c.E.SeedAndFilter(vec, False, 0), want)


@pytest.mark.parametrize("max_hits,size", [(5000, 80000), (700, 50000), (150, 24000)])
def test_max_hits_iteration_split(oracle, engine_clean, max_hits, size):
    """Hazard H4: per-iteration dedup; small MAX_HITS -> many iterations (more than one extension batch of 8 segments;
    hundreds of iterations per call in the last case, sized so that it still runs in seconds)."""
    t, q = synth.make_pair(size, 5, 6, sub_rate=0.08, mask_frac=0.05, records=2)
    c = Case(t, q, chunk=size // 2).oracle_setup(oracle).engine_setup(engine_clean)
    c.E.set_max_hits(max_hits)
    assert c.E.get_max_hits() == max_hits
    n = run_all_chunks(c, max_hits=max_hits)
    assert n > 0
    st = c.E.last_call_stats()
    assert st["num_iter"] >= 2


@pytest.mark.parametrize("shape,step,transition", [
    (SHAPE_12OF19, 2, True), (SHAPE_12OF19, 3, False), (SHAPE_12OF19, 4, True),
    ("T0T0TT00T0TTT", 1, True),              # custom pattern, weight 8 (main.cpp:168-178)
    ("TTTTT11TTTT", 1, True),                # '1' care positions are not transition-enabled (ntcoding.cpp:21-37)
    ("TTT0T0TT00TT00T0T0TTTT", 1, False),    # 14of22 (main.cpp:164-167): 4^14 buckets, transitions off
    ("TTT0T0TT00TT00T0T0TTTT", 1, True),     # 14of22 as the reference runs it: 15 seed words per position
])
def test_shapes_steps_and_transitions(oracle, engine_clean, shape, step, transition):
    t, q = synth.make_pair(90000, 7, 8, sub_rate=0.08, mask_frac=0.1, records=3, n_runs=2)
    c = Case(t, q, shape=shape, step=step, transition=transition, chunk=45000).oracle_setup(oracle).engine_setup(engine_clean)
    assert np.array_equal(c.E.copy_index_table(), c.o_index)
    assert np.array_equal(c.E.copy_pos_table(), c.o_pos)
    per = 1 + (bin(int("".join("1" if ch == "T" else "0" for ch in shape if ch in "T1")[::-1] or "0", 2)).count("1") if transition else 0)
    for rev in (False, True):
        (s, e) = c.chunks()[0]
        assert np.array_equal(c.E.device_make_seeds(s, e, rev, 0, per=max(per, 1)), c.host_seeds(s, e, rev))
    assert run_all_chunks(c) > 0


@pytest.mark.parametrize("xdrop,hspthresh,noentropy", [(910, 3000, True), (910, 2200, False), (300, 1500, False),
                                                       (0, 1200, False), (5000, 6000, False)])
def test_scoring_parameters(oracle, engine_clean, xdrop, hspthresh, noentropy):
    """xdrop 300/0 -> 7*max(M) > xdrop: the filter must use its exact (sticky) path; 2200 -> many entropy candidates."""
    t, q = synth.make_pair(70000, 9, 10, sub_rate=0.12, mask_frac=0.1, indel_every=150)
    c = Case(t, q, xdrop=xdrop, hspthresh=hspthresh, noentropy=noentropy, chunk=35000)
    c.oracle_setup(oracle).engine_setup(engine_clean)
    assert run_all_chunks(c) > 0


def test_low_complexity_entropy_branch(oracle, engine_clean):
    """Poly-A / CT-repeat homology: scores inside [hspthresh, 3*hspthresh] whose entropy factor decides."""
    rng = np.random.default_rng(11)
    t = synth.random_dna(60000, 11)
    q = synth.random_dna(60000, 12)
    for i in range(40):
        ln = int(rng.integers(40, 120))
        a, b = int(rng.integers(100, 59000)), int(rng.integers(100, 59000))
        if i % 3 == 0:
            seg = np.where(rng.random(ln) < 0.85, ord("A"), synth.random_dna(ln, 100 + i))
        elif i % 3 == 1:
            seg = np.tile(np.frombuffer(b"CT", dtype=np.uint8), ln // 2 + 1)[:ln]
        else:
            seg = synth.random_dna(ln, 200 + i)
        t[a:a + ln] = seg
        q[b:b + ln] = synth.mutate(seg.astype(np.uint8), 300 + i, 0.03)
    c = Case(t, q, chunk=60000).oracle_setup(oracle).engine_setup(engine_clean)
    n = run_all_chunks(c)
    seeds = c.host_seeds(0, q.size - 19, False)
    c.E.SeedAndFilter(seeds, False, 0)
    assert c.E.last_call_stats()["num_entropy"] > 0 and n > 0


def test_tiny_and_ragged_blocks(oracle, engine_clean):
    """Blocks barely longer than the seed, records of wildly different lengths, N runs and '&' at the edges."""
    recs = [synth.random_dna(n, 20 + i) for i, n in enumerate((19, 20, 45, 3000, 64, 7000, 21))]
    t = synth.join_records(recs)
    q = synth.join_records([synth.mutate(r, 50 + i, 0.05) for i, r in enumerate(recs[::-1])])
    q[100:140] = ord("N")
    c = Case(t, q, chunk=2500).oracle_setup(oracle).engine_setup(engine_clean)
    assert np.array_equal(c.E.copy_pos_table(), c.o_pos)
    run_all_chunks(c)
    # a 30-base target against a 25-base query
    c2 = Case(synth.random_dna(30, 1), synth.random_dna(25, 1), chunk=100).oracle_setup(oracle)
    engine_clean.ShutdownProcessor()
    c2.engine_setup(engine_clean)
    run_all_chunks(c2)


def test_concurrent_host_threads_share_the_token_pool(oracle, engine_clean):
    """Up to num_threads TBB workers call g_SeedAndFilter concurrently in the reference."""
    t, q = synth.make_pair(200000, 13, 14, sub_rate=0.1, mask_frac=0.1, records=2)
    c = Case(t, q, chunk=25000).oracle_setup(oracle).engine_setup(engine_clean)
    jobs = [(rev, s, e) for rev in (False, True) for (s, e) in c.chunks()]
    want = {j: c.oracle_saf(c.host_seeds(j[1], j[2], j[0]), j[0])[0] for j in jobs}
    got = {}
    lock = threading.Lock()

    def work(my):
        for j in my:
            out = c.E.SeedAndFilter(c.host_seeds(j[1], j[2], j[0]), j[0], 0)
            with lock:
                got[j] = out

    threads = [threading.Thread(target=work, args=(jobs[i::4],)) for i in range(4)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert all(seg_equal(got[j], want[j]) for j in jobs)


def test_repeat_masker_variant(oracle, engine_clean):
    """repeat_masker_src/seed_filter.cu: query = the target itself, window filter, rc coordinate flip, 5-step chain,
    64-bit header (SURVEY a-10)."""
    unit = synth.random_dna(400, 77)
    t = synth.random_dna(120000, 15)
    rng = np.random.default_rng(3)
    for i in range(60):  # plant a diverged repeat family
        p = int(rng.integers(0, t.size - 500))
        cp = synth.mutate(unit, 500 + i, 0.06)
        t[p:p + cp.size] = cp if i % 4 else synth.reverse_complement(cp)
    t = synth.soft_mask(t, 5, 0.05)
    c = Case(t, t, chunk=30000).oracle_setup(oracle).engine_setup(engine_clean)
    E, O = c.E, oracle
    E.RmSendQueryWriteRequest()
    t_rc_ascii = np.frombuffer(O.rev_comp_ascii(t.tobytes(), 0, t.size), dtype=np.uint8)
    o_rc = O.rev_comp_codes(c.o_ref)
    assert np.array_equal(O.encode(t_rc_ascii.tobytes()), o_rc)
    total = 0
    for rev in (False, True):
        buf = t_rc_ascii if rev else t
        qcodes = o_rc if rev else c.o_ref
        for (s, e) in c.chunks():
            seeds = O.make_seeds(buf.tobytes(), 0, s, e, 19, c.kmer_size, True)
            for (ws, we) in ((0, t.size), (s, min(s + 50000, t.size)), (1000, 900)):
                want, st = O.seed_and_filter(c.o_ref, qcodes, c.o_index, c.o_pos, seeds, c.sub_mat, rm=(rev, ws, we))
                got = E.RmSeedAndFilter(seeds, rev, ws, we)
                assert seg_equal(got, want), (rev, s, e, ws, we, got[:3], want[:3])
                total += want.size - 1
    assert total > 0
    E.RmClearQuery()


@pytest.mark.parametrize("atomic", [0, 1])
@pytest.mark.parametrize("shape,step", [(SHAPE_12OF19, 1), (SHAPE_12OF19, 3), ("TTTTT11TTTT", 1), ("TT0TT0TT0TT0T", 2)])
def test_both_table_builds_give_the_reference_table(oracle, engine_clean, atomic, shape, step):
    """GenerateSeedPosTable (common/seed_pos_table.cu:49-109) on the device, both builds: the LDS-staged partition build (seed
    weights 9..12, the default) and the atomic counting sort (option table_atomic, and every other weight) must give the
    reference's index table and -- canonically ordered -- position table, incl. masked runs, N runs, record separators, a
    poly-A stretch that makes one coarse partition overflow the finishing kernel's LDS, and steps > 1."""
    t, q = synth.make_pair(3_000_000, 17, 18, sub_rate=0.08, mask_frac=0.15, records=4, n_runs=3)
    t = t.copy()
    t[1_200_000:1_260_000] = ord("A")  # 60 k positions with ONE key: a bucket (and a coarse partition) far above every LDS capacity
    engine_clean.set_option("table_atomic", atomic)
    try:
        c = Case(t, q[:200000], shape=shape, step=step, chunk=100000).oracle_setup(oracle).engine_setup(engine_clean)
        assert np.array_equal(c.E.copy_index_table(), c.o_index)
        assert np.array_equal(canonical_pos_table(c.o_index, c.E.copy_pos_table()), canonical_pos_table(c.o_index, c.o_pos))
        # buckets up to the sort limit are in ascending order on the device as well (hazard H7 is about the rest)
        idx = c.o_index.astype(np.int64)
        starts = np.concatenate([[0], idx[:-1]])
        small = np.flatnonzero((idx - starts >= 2) & (idx - starts <= 512))[:20000]
        pos = c.E.copy_pos_table()
        for k in small[:: max(1, small.size // 500)]:
            seg = pos[starts[k]:idx[k]]
            assert np.all(seg[1:] > seg[:-1]), int(k)
        run_all_chunks(c, rev_list=(False,))
    finally:
        engine_clean.reset_option("table_atomic")
