"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_case(oracle, engine):
    t, q = synth.make_pair(300000, 11, 12, sub_rate=0.10, mask_frac=0.1, records=3, indel_every=400, n_runs=2)
    c = Case(t, q, chunk=100000).oracle_setup(oracle).engine_setup(engine)
    yield c
    engine.ShutdownProcessor()


def test_encode_matches_oracle(small_case):
    c = small_case
    assert np.array_equal(c.E.copy_ref_codes(), c.o_ref)
    assert np.array_equal(c.E.copy_query_codes(0, False), c.o_q)
    assert np.array_equal(c.E.copy_query_codes(0, True), c.o_qrc)


def test_seed_pos_table_matches_oracle(small_case):
    c = small_case
    assert np.array_equal(c.E.copy_index_table(), c.o_index)
    assert np.array_equal(c.E.copy_pos_table(), c.o_pos)


def test_device_seeder_matches_host_loop(small_case):
    c = small_case
    for rev in (False, True):
        for (s, e) in c.chunks():
            assert np.array_equal(c.E.device_make_seeds(s, e, rev, 0), c.host_seeds(s, e, rev))


@pytest.mark.parametrize("rev", [False, True])
def test_seed_and_filter_bit_exact(small_case, rev):
    c = small_case
    total = 0
    for (s, e) in c.chunks():
        seeds = c.host_seeds(s, e, rev)
        got = c.E.SeedAndFilter(seeds, rev, 0)
        want, st = c.oracle_saf(seeds, rev)
        assert seg_equal(got, want), (s, e, got[:4], want[:4])
        got2 = c.E.SeedAndFilterRange(s, e, rev, 0)
        assert seg_equal(got2, want)
        total += want.size - 1
    assert total > 0


def test_examined_counter_matches_oracle(small_case):
    """E of SURVEY 8(d) -- positions the reference algorithm scores -- counted on the device must equal the oracle's
    count (it feeds roofline.achieved), and counting must not change the result."""
    c = small_case
    (s, e) = c.chunks()[0]
    seeds = c.host_seeds(s, e, False)
    want, st = c.oracle_saf(seeds, False)
    c.E.set_count_examined(True)
    try:
        got = c.E.SeedAndFilter(seeds, False, 0)
        stats = c.E.last_call_stats()
    finally:
        c.E.set_count_examined(False)
    assert seg_equal(got, want)
    assert stats["num_hits"] == st["num_hits"]
    assert stats["num_survivors"] == st["num_survivors"]
    assert stats["num_examined"] == st["num_examined"]
    assert stats["num_candidates"] >= stats["num_survivors"]
    assert stats["num_examined_filter"] > 0


@pytest.mark.parametrize("threads", [1, 3])
def test_seed_interval_equals_the_chunk_loop(small_case, threads):
    """sa_seed_interval = seeder_body::operator() (src/seeder.cpp:12-127): plus-strand chunks, then minus-strand chunks in
    rc coordinates, HSPs concatenated per strand without headers -- against the oracle chunk by chunk."""
    c = small_case
    from segalign_amd import shard
    q_len = c.query.size - c.seed_size
    iv = (30000, min(260000, q_len))
    want = {False: [], True: []}
    hits = 0
    for rev in (False, True):
        for (s, e) in shard.chunks_of(iv, c.chunk, q_len, rev):
            segs, st = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            want[rev].append(segs[1:])
            hits += st["num_hits"]
    fw, rc, tot = c.E.SeedInterval(iv[0], iv[1], q_len, c.E.STRAND_BOTH, 0, threads)
    assert seg_equal(fw, np.concatenate(want[False])) and seg_equal(rc, np.concatenate(want[True]))
    assert tot["num_hits"] == hits and tot["num_anchors"] == fw.size + rc.size
    fw1, rc1, _ = c.E.SeedInterval(iv[0], iv[1], q_len, c.E.STRAND_MINUS, 0, threads)
    assert fw1.size == 0 and seg_equal(rc1, rc)


@pytest.mark.parametrize("rev", [False, True])
def test_multi_chunk_call_equals_one_call_per_chunk(small_case, rev):
    """sa_seed_and_filter_chunks: several chunks share one pass over the kernels, but every chunk keeps its own iteration
    plan, dedup scope and header -- the vectors must equal the per-chunk calls (and the oracle) exactly"""
    c = small_case
    q_len = c.query.size - c.seed_size
    k = 3
    for start in (0, 40000):
        end = min(start + k * c.chunk, q_len)
        got = c.E.SeedAndFilterChunks(start, end, rev, 0)
        nch = -(-(end - start) // c.chunk)
        assert all(g.size == 0 for g in got[nch:])
        for i in range(nch):
            s, e = start + i * c.chunk, min(start + (i + 1) * c.chunk, end)
            want, _ = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            assert seg_equal(got[i], want), (rev, start, i, got[i][:3], want[:3])
            assert seg_equal(got[i], c.E.SeedAndFilterRange(s, e, rev, 0))


def test_interval_entry_equals_drop_in_path_on_the_tail_interval(small_case):
    """The reference seeder is handed q_len = block length - seed size (src/main.cpp:708), so on both strands every seed
    window of the LAST interval ends inside the block (j + 18 <= len - 2).  The additive entries clamp seed starts to
    j + span <= len; on the reference's own call pattern that clamp never binds: the interval entry equals the drop-in
    path chunk by chunk, tail included."""
    c, E = small_case, small_case.E
    q_len = c.query.size - c.seed_size
    iv = (q_len - 150000, q_len)  # the tail interval (1.5 chunks of 100 kbp)
    fw, rc, st = E.SeedInterval(iv[0], iv[1], q_len, E.STRAND_BOTH, 0, 2)
    for rev, got in ((False, fw), (True, rc)):
        a, b = (iv[0], iv[1]) if not rev else (q_len - iv[1], q_len - iv[0])
        parts = []
        for s in range(a, b, c.chunk):
            seeds = c.host_seeds(s, min(s + c.chunk, b), rev)
            if seeds.size:
                parts.append(E.SeedAndFilter(seeds, rev, 0)[1:])
        want = np.concatenate(parts) if parts else np.zeros(0, dtype=fw.dtype)
        assert np.array_equal(got, want), rev
