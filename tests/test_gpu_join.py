"""GPU: KEY-ORDERED calls (segalign_amd/csrc/join.hip, extend.hip 1e) give the oracle's vectors bit for bit.

A key-ordered call sorts its query positions by seed key and enumerates the hits per key -- (run of context records) x (positions
that carry the key) -- instead of in query order; what the reference fixes by hit ORDER (the iteration split of
src/seed_filter.cu:718-745, hence the dedup scopes of :776-782) is resolved per hit from (position, entry index inside the run).
Forced here with option key_order = 2 (by default only calls with about a position per seed key or more take this form), on inputs that
reach its corners: chunks without seeds, soft-masked chunks, a short last chunk, a query that repeats one stretch forty times (keys with
more than 16 positions: several entries per key), poly-A (runs of tens of thousands of records: tiles inside one run), tiny runs
(--notransition: tiles that span dozens of entries), both strands, MAX_HITS splits (the call halves itself until the chunk in
question runs alone), list regrowth, and the audit of every rejected hit."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu

_engine = None


def with_env(env):
    """options of the next InitializeProcessor: {"SEGALIGN_AMD_KEY_ORDER": "2"} -> option key_order = 2"""
    _engine.reset_option(None)
    for k, v in env.items():
        _engine.set_option(k[len("SEGALIGN_AMD_"):].lower(), int(v))


@pytest.fixture
def clean(engine):
    global _engine
    _engine = engine
    yield engine
    engine.ShutdownProcessor()
    engine.set_max_hits(0)
    engine.reset_option(None)


def run_calls(c, groups, key_order, rev):
    """-> per-chunk vectors of the chunks of `groups` ([(first chunk, last chunk)]) through multi-chunk calls"""
    ch = c.chunks()
    outs, flags = [], 0
    for (g0, g1) in groups:
        o = c.E.SeedAndFilterChunks(ch[g0][0], ch[g1][1], rev, 0)  # (one slot per chunk the engine can carry: the call's chunks come first)
        flags |= c.E.last_call_stats()["path_flags"]
        outs.extend(o[:g1 - g0 + 1])
    return outs, flags


def check_case(oracle, engine, t, q, chunk, groups=None, transition=True, env=None, **kw):
    with_env(dict({"SEGALIGN_AMD_KEY_ORDER": "2"}, **(env or {})))
    c = Case(t, q, chunk=chunk, transition=transition, **kw).oracle_setup(oracle).engine_setup(engine)
    ch = c.chunks()
    groups = groups or [(0, len(ch) - 1)]
    total = 0
    check_case.flags = 0
    for rev in (False, True):
        wants = []
        for (g0, g1) in groups:
            wants.extend(c.oracle_saf(c.host_seeds(s, e, rev), rev)[0] for (s, e) in ch[g0:g1 + 1])
        outs, flags = run_calls(c, groups, 2, rev)
        check_case.flags |= flags
        assert flags & engine.PATH_KEY_ORDERED, "the call did not go key-ordered"
        assert len(outs) == len(wants)
        for j, (o, w) in enumerate(zip(outs, wants)):
            if w.size <= 1 and (o is None or o.size == 0):
                continue  # (a chunk without seeds returns nothing, seeder.cpp:76)
            assert seg_equal(o, w), (rev, j, None if o is None else o.size, w.size)
            total += w.size - 1
    return c, total


@pytest.mark.parametrize("transition", [True, False])
def test_key_ordered_calls_equal_the_oracle(oracle, clean, transition):
    t, q = synth.make_pair(400000, 41, 42, sub_rate=0.09, mask_frac=0.15, records=3, indel_every=450, n_runs=2)
    c, total = check_case(oracle, clean, t, q, 24000, groups=[(0, 15), (16, 16), (3, 9)], transition=transition)  # 17 chunks, the last one short
    assert total > 100


def test_key_ordered_equals_streamed_on_every_chunk_and_strand(oracle, clean):
    """the same multi-chunk calls, streamed (key_order = 0) and key-ordered: identical vectors, identical hit counts"""
    t, q = synth.make_pair(600000, 11, 12, sub_rate=0.07, mask_frac=0.2, records=4, indel_every=700, invert_frac=0.3, invert_block=20000)
    res = {}
    for ko in ("0", "2"):
        with_env({"SEGALIGN_AMD_KEY_ORDER": ko})
        c = Case(t, q, chunk=30000).oracle_setup(oracle).engine_setup(clean)
        ch = c.chunks()
        res[ko] = []
        for rev in (False, True):
            outs = c.E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)[:len(ch)]
            st = c.E.last_call_stats()
            assert bool(st["path_flags"] & clean.PATH_KEY_ORDERED) == (ko == "2")
            res[ko].append((outs, st["num_hits"], st["num_seeds"]))
        clean.ShutdownProcessor()
    for a, b in zip(res["0"], res["2"]):
        assert a[1] == b[1] and a[2] == b[2]
        assert len(a[0]) == len(b[0]) and all(seg_equal(x, y) for x, y in zip(a[0], b[0]))
    assert sum(o.size for o in res["2"][0][0]) > 50


def test_empty_masked_and_repeated_chunks(oracle, clean):
    t, q = synth.make_pair(300000, 61, 62, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=600)
    q = q.copy()
    chunk = 20000
    q[3 * chunk:5 * chunk] = ord("N")                      # chunks 3, 4: no seeds at all
    q[9 * chunk:10 * chunk] = np.frombuffer(bytes(q[9 * chunk:10 * chunk]).lower(), dtype=np.uint8)  # chunk 9: soft-masked
    # chunks 11, 12: one 1 kb stretch of the target forty times over -- every key of it has 40 positions in the call (> 16: three entries)
    unit = t[150000:151000]
    q[11 * chunk:11 * chunk + 40 * unit.size] = np.tile(unit, 40)
    c, total = check_case(oracle, clean, t, q, chunk)
    assert total > 200


def test_heavy_keys_poly_a_and_microsatellites(oracle, clean):
    """runs far longer than a tile (poly-A: one key holds thousands of records, and the query has thousands of positions with it)"""
    rng = np.random.default_rng(5)
    t = synth.random_dna(200000, 81)
    q = synth.mutate(t, 82, 0.06)
    t, q = t.copy(), q.copy()
    t[40000:43000] = ord("A")
    t[90000:92000] = np.frombuffer(b"AC" * 1000, dtype=np.uint8)
    q[60000:62500] = ord("A")
    q[120000:121000] = np.frombuffer(b"CA" * 500, dtype=np.uint8)
    c, total = check_case(oracle, clean, t, q, 25000, hspthresh=2200)
    assert total > 100


def test_a_max_hits_split_sends_the_chunk_down_the_general_path(oracle, clean):
    with_env({"SEGALIGN_AMD_KEY_ORDER": "2"})
    t, q = synth.make_pair(100000, 51, 52, sub_rate=0.07, mask_frac=0.1)
    c = Case(t, q, chunk=12500).oracle_setup(oracle).engine_setup(clean)
    ch = c.chunks()
    for mh in (30000, 1 << 30):
        c.E.set_max_hits(mh)
        for rev in (False, True):
            wants = [c.oracle_saf(c.host_seeds(s, e, rev), rev, max_hits=mh)[0] for (s, e) in ch]
            outs = c.E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)
            for j, (o, w) in enumerate(zip(outs, wants)):
                assert seg_equal(o, w), (mh, rev, j)
    c.E.set_max_hits(0)


def test_a_call_far_above_its_hit_bound_is_halved(oracle, clean):
    """Calls are sized by the hits random sequence would collect; a call that turns out to hold more than 1.5 x option key_order_hits
    is halved (front.hip join_front) -- down to single chunks, which stay key-ordered -- with the same vectors."""
    t, q = synth.make_pair(300000, 33, 34, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=500)
    c, total = check_case(oracle, clean, t, q, 20000, env={"SEGALIGN_AMD_KEY_ORDER_HITS": str(1 << 20)})  # ~1.5 M hits per chunk: every multi-chunk pass splits
    assert total > 100


def test_list_regrowth_reruns_the_key_ordered_filter(oracle, clean):
    t, q = synth.make_pair(200000, 71, 72, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=500)
    c, total = check_case(oracle, clean, t, q, 12000, env={"SEGALIGN_AMD_L2_CAP": "1024"})
    assert total > 64 and check_case.flags & clean.PATH_LIST_REGROWN  # (the first call regrows the slot's list for good)


def test_every_rejected_hit_really_fails(oracle, clean):
    """the audit of tests/test_gpu_filter_audit.py on a key-ordered call: every hit the two filter levels reject is extended exactly by
    the oracle and must not pass"""
    with_env({"SEGALIGN_AMD_KEY_ORDER": "2", "SEGALIGN_AMD_AUDIT_CAP": str(1 << 22)})
    t, q = synth.make_pair(60000, 91, 92, sub_rate=0.10, mask_frac=0.1, indel_every=400)
    c = Case(t, q, chunk=15000).oracle_setup(oracle).engine_setup(clean)
    ch = c.chunks()
    for rev in (False, True):
        outs = c.E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)
        st = c.E.last_call_stats()
        assert st["path_flags"] & clean.PATH_KEY_ORDERED
        wants = [c.oracle_saf(c.host_seeds(s, e, rev), rev)[0] for (s, e) in ch]
        assert all(seg_equal(o, w) for o, w in zip(outs, wants))
        pairs, n = c.E.get_audit()
        assert n == pairs.shape[0] and n > 1000, "audit list overflowed or empty"
        assert n + st["num_candidates"] <= st["num_hits"]
        qcodes = c.o_qrc if rev else c.o_q
        ok, recs = oracle.extend_hits_pass(c.o_ref, qcodes, c.sub_mat, pairs, xdrop=c.xdrop, hspthresh=c.hspthresh, noentropy=c.noentropy)
        assert np.count_nonzero(ok) == 0
