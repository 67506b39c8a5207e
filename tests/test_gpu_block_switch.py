"""GPU: the reference's block loop at size -- g_ClearRef -> next target block -> BUFFER_DEPTH query blocks swapping
(src/main.cpp:601-685) -- with target blocks large -> small -> large, so the engine's table arena is filled, partly reused and
grown again, and a block whose context table does not fit next to the work buffers falls back to the position-only table
(lookup mode 1) and comes back to mode 2 on the next block.  Per block: lookup mode, table footprint, and two chunks per query
buffer and strand bit-exact against the oracle (the oracle runs on the table copied from the device: building a 200 Mbp table
on the CPU takes longer than the whole GPU suite)."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import shard, synth

pytestmark = pytest.mark.gpu

SHAPE = "TTT0T00TT00T0T0TTTT"
CHUNK = 250000


def make_block(n, seed):
    t = synth.soft_mask(synth.random_dna(n, seed), seed + 1, 0.25, 200, 2000)
    per = n // 3
    return synth.join_records([t[i * per:(i + 1) * per] for i in range(3)])


def make_query(target, seed, pieces=4, piece=260000):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(pieces):
        p = int(rng.integers(0, target.size - piece - 1))
        seg = synth.mutate(target[p:p + piece].copy(), seed + 10 + i, 0.03, indel_every=700)
        out.append(synth.reverse_complement(seg) if i % 2 else seg)
    return np.concatenate(out)


def check_block(E, O, sub_mat, k, target, queries, expect_mode):
    """one pass of the reader lambda for a target block (main.cpp:613-621,649-664) + chunk calls on both buffers"""
    keep = E.SendRefWriteRequest(target, 0, target.size)
    E.GenerateSeedPosTable(keep, 0, target.size, 1, 19, k)
    for b, q in enumerate(queries):
        E.SendQueryWriteRequest(q, 0, q.size, b)
    assert E.lookup_mode() == expect_mode
    index, pos, rcodes = E.copy_index_table(), E.copy_pos_table(), E.copy_ref_codes()
    if expect_mode:
        assert E.neighbourhood_entries() == 13 * pos.size
    hsps = 0
    for b, q in enumerate(queries):
        for rev in (False, True):
            qcodes = E.copy_query_codes(b, rev)
            buf = q if not rev else np.frombuffer(O.rev_comp_ascii(q.tobytes(), 0, q.size), dtype=np.uint8)
            for a in (0, CHUNK):
                e = min(a + CHUNK, q.size - 19)
                seeds = O.make_seeds(buf.tobytes(), 0, a, e, 19, k, True)
                want, st = O.seed_and_filter(rcodes, qcodes, index, pos, seeds, sub_mat)
                got = E.SeedAndFilterRange(a, e, rev, b)
                assert got.shape == want.shape and np.all(got == want), (target.size, b, rev, a)
                assert E.last_call_stats()["lookup_path"] == expect_mode
                got2 = E.SeedAndFilter(seeds, rev, b)  # the drop-in entry on the same chunk
                assert got2.shape == want.shape and np.all(got2 == want), (target.size, b, rev, a, "drop-in")
                hsps += want.size - 1
    for b in range(len(queries)):
        E.ClearQuery(b)  # main.cpp:659,680
    return hsps


def test_large_small_large_blocks_through_clear_ref(oracle, engine):
    E, O = engine, oracle
    sub_mat = O.build_sub_mat(910)
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    O.generate_shape_pos(SHAPE)
    E.InitializeProcessor(True, CHUNK, 19, sub_mat, 910, 3000, False)
    try:
        big = make_block(200_000_000, 11)
        small = make_block(50_000_000, 21)
        total = 0
        for i, t in enumerate((big, small, big)):
            if i:
                E.ClearRef()  # main.cpp:613
            total += check_block(E, O, sub_mat, k, t, [make_query(t, 100 + 7 * i), make_query(t, 200 + 7 * i)], 2)
        assert total > 100
    finally:
        E.ShutdownProcessor()


def test_block_that_does_not_fit_falls_back_to_positions_and_recovers(oracle, engine):
    """Free HBM is taken away (a foreign allocation) between two blocks, so the context table of the second, larger block does not
    fit next to the engine's reserve: that block runs with the position-only table (mode 1) and gives the arena back; once the
    memory is free again the next block is back in mode 2.  Results equal the oracle in every mode."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the HIP runtime the engine itself is linked against (same process-wide instance)
    E, O = engine, oracle
    sub_mat = O.build_sub_mat(910)
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    O.generate_shape_pos(SHAPE)
    # the table arena is a process-wide cache: give back what earlier tests left mapped (arena_gb = 0 releases it at
    # ShutdownProcessor), then run with a small arena, so that what is mapped when the memory is taken away is known
    # (and no work arena: its background mapping would take memory while this test does its own accounting)
    E.set_option("arena_gb", 0)
    E.set_option("work_gb", 0)
    E.InitializeProcessor(True, CHUNK, 19, sub_mat, 910, 3000, False)
    E.ShutdownProcessor()
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    E.set_option("arena_gb", 8)
    E.InitializeProcessor(True, CHUNK, 19, sub_mat, 910, 3000, False)
    hog = None
    try:
        small = make_block(20_000_000, 31)
        big = make_block(160_000_000, 41)
        check_block(E, O, sub_mat, k, small, [make_query(small, 300)], 2)
        free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
        hog = ctypes.c_void_p()  # leave ~30 GB: the table (~52 GB) + the engine's reserve (24 GiB) exceed that + the arena (8 GiB),
        assert hip.hipMalloc(ctypes.byref(hog), ctypes.c_size_t(max(free_b.value - (30 << 30), 1 << 20))) == 0  # positions + reserve fit
        E.ClearRef()
        check_block(E, O, sub_mat, k, big, [make_query(big, 400)], 1)
        assert hip.hipFree(hog) == 0
        hog = None
        E.ClearRef()
        check_block(E, O, sub_mat, k, big, [make_query(big, 500)], 2)
    finally:
        if hog is not None:
            hip.hipFree(hog)
        E.ShutdownProcessor()
        E.reset_option("arena_gb")
        E.reset_option("work_gb")


def test_work_regions_survive_a_reinitialisation_with_another_layout(oracle, engine):
    """InitializeProcessor again WITHOUT a ShutdownProcessor in between, with another work-arena share per slot (option work_gb) and
    another slot count: slots that exist already are torn down and set up again before they are re-based (round 4's advisor: their
    buffers pointed into the old layout, where a neighbour's new region carved over them).  Multi-chunk calls on several host
    threads before and after, against the oracle."""
    E = engine
    t, q = synth.make_pair(600000, 71, 72, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=600)
    c = Case(t, q, chunk=50000)
    c.oracle_setup(oracle)
    want = {}
    q_len = q.size - 19
    for rev in (False, True):
        exp = []
        for (s, e) in shard.chunks_of((0, q_len), 50000, q_len, rev):
            seeds = c.host_seeds(s, e, rev)
            if seeds.size:
                exp.append(c.oracle_saf(seeds, rev)[0][1:])
        want[rev] = np.concatenate(exp)
    try:
        for (work_gb, slots) in ((3, 4), (1, 6), (2, 2), (0, 4)):
            E.set_option("work_gb", work_gb)
            E.set_option("slots", slots)
            if work_gb == 3:
                c.engine_setup(E)
            else:   # the reference's order after a parameter change: processor, target, table, query -- no shutdown
                E.InitializeProcessor(True, 50000, 19, c.sub_mat, 910, 3000, False)
                keep = E.SendRefWriteRequest(t, 0, t.size)
                E.GenerateSeedPosTable(keep, 0, t.size, 1, 19, c.kmer_size)
                E.SendQueryWriteRequest(q, 0, q.size, 0)
            for rep in range(2):
                fw, rc, tot = E.SeedInterval(0, q_len, q_len, E.STRAND_BOTH, 0, 3)
                assert seg_equal(fw, want[False]) and seg_equal(rc, want[True]), (work_gb, slots, rep)
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)


@pytest.mark.parametrize("frees", [0, 1])
def test_clear_ref_keeps_or_frees_the_table_buffers(oracle, engine, frees):
    """g_ClearRef between target blocks of growing and shrinking size: by default the engine forgets the tables and keeps their buffers
    (grown before they are re-allocated: freed first), with option clear_ref_frees = 1 it frees them like the reference's clearRef
    (common/seed_filter_interface.cu:103-113).  Either way every block's tables are the oracle's and its chunk calls bit-exact."""
    E, O = engine, oracle
    E.set_option("clear_ref_frees", frees)
    try:
        first = True
        for n, seed in ((600_000, 301), (2_400_000, 302), (900_000, 303)):
            t, q = synth.make_pair(n, seed, seed + 50, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=600)
            c = Case(t, q, chunk=100_000).oracle_setup(O)
            if first:
                c.engine_setup(E)
                assert E.get_option("clear_ref_frees") == frees
                first = False
            else:  # the reader lambda's order for a later block: main.cpp:613-621,659-661
                E.ClearRef()
                E.ClearQuery(0)
                c.E = E
                keep = E.SendRefWriteRequest(c.target, 0, c.target.size)
                E.GenerateSeedPosTable(keep, 0, c.target.size, 1, 19, c.kmer_size)
                E.SendQueryWriteRequest(c.query, 0, c.query.size, 0)
            assert np.array_equal(E.copy_index_table(), c.o_index) and np.array_equal(E.copy_pos_table(), c.o_pos)
            n_hsps = 0
            for rev in (False, True):
                for (s, e) in c.chunks()[:4]:
                    want, _ = c.oracle_saf(c.host_seeds(s, e, rev), rev)
                    assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), want), (frees, n, rev, s)
                    n_hsps += want.size - 1
            assert n_hsps > 0
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)
