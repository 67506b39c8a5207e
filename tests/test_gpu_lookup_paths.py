"""GPU: the three seed-lookup paths of the device-seeded entry points give the oracle's result bit for bit:
  2  table-direct with target context (neighbourhood table of 28-byte records, class filter + second level; default)
  1  table-direct, positions only (SEGALIGN_AMD_NO_CTX=1; what a target block too large for the context table gets)
  0  general path: seed words -> bucket lookup -> hit list (SEGALIGN_AMD_NO_TD=1; also every drop-in call)
src/seed_filter.cu:157-230 + :682-828 (lookup, iteration plan, hits) on every path; repeat-masker variant included."""
import os

import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth
from test_gpu_rm_mask import as_list, model_mask_interval

pytestmark = pytest.mark.gpu

MODES = [(2, {}), (1, {"SEGALIGN_AMD_NO_CTX": "1"}), (0, {"SEGALIGN_AMD_NO_TD": "1"})]


def with_env(env):
    for k in ("SEGALIGN_AMD_NO_CTX", "SEGALIGN_AMD_NO_TD", "SEGALIGN_AMD_SPEC_RECS", "SEGALIGN_AMD_DEDUP_SEG_MAX",
              "SEGALIGN_AMD_SPEC_DEDUP", "SEGALIGN_AMD_NO_SMALL_DEDUP", "SEGALIGN_AMD_L2_CAP"):
        os.environ.pop(k, None)
    os.environ.update(env)


@pytest.fixture
def clean(engine):
    yield engine
    engine.ShutdownProcessor()
    engine.set_max_hits(0)
    with_env({})


@pytest.mark.parametrize("mode,env", MODES)
@pytest.mark.parametrize("transition", [True, False])
def test_every_lookup_path_equals_the_oracle(oracle, clean, mode, env, transition):
    with_env(env)
    t, q = synth.make_pair(400000, 41, 42, sub_rate=0.09, mask_frac=0.15, records=3, indel_every=450, n_runs=2)
    c = Case(t, q, chunk=24000, transition=transition).oracle_setup(oracle).engine_setup(clean)  # 17 chunks: a full 16-chunk call + 1
    E = c.E
    assert E.lookup_mode() == mode
    if mode:
        words = 13 if transition else 1
        assert E.neighbourhood_entries() == words * c.o_pos.size  # every bucket appears once per seed word that maps to it
    q_len = q.size - c.seed_size
    per = {False: [], True: []}
    for rev in (False, True):
        wants = []
        for (s, e) in c.chunks():
            want, st = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), want), (mode, rev, s, e)
            assert E.last_call_stats()["num_hits"] == st["num_hits"]
            wants.append(want)
            per[rev].append(want[1:])
        ch = c.chunks()
        kmax = E.lib().sa_max_chunks_per_call()
        assert kmax == 256 and len(ch) == 17
        for k in (4, 16, kmax):  # multi-chunk calls: own plan / dedup scope / vector per chunk (all 17 chunks in one call = 34 reference iterations)
            for g in range(0, len(ch), k):
                outs = E.SeedAndFilterChunks(ch[g][0], ch[min(g + k - 1, len(ch) - 1)][1], rev, 0)
                for j, w in enumerate(wants[g:g + k]):
                    assert seg_equal(outs[j], w), (mode, rev, k, g, j)
    fw, rc, st = E.SeedInterval(0, q_len, q_len, E.STRAND_BOTH, 0, 3)
    assert np.array_equal(fw, np.concatenate(per[False])) and np.array_equal(rc, np.concatenate(per[True]))
    assert fw.size + rc.size > 100


@pytest.mark.parametrize("mode,env", MODES)
def test_lookup_paths_under_a_max_hits_split(oracle, clean, mode, env):
    """num_hits >= MAX_HITS (hazard H4): more than two reference iterations.  A table-direct call plans the reference's greedy groups
    (src/seed_filter.cu:725-741) itself from the position records and the plain bucket sizes (probe.hip probe_plan_kernel) as long as a
    chunk needs at most TD_MAX_ITER = 8 of them; beyond that it hands the chunk to the general path (per-seed-word prefixes); below the
    limit it keeps the two-iteration split."""
    with_env(env)
    t, q = synth.make_pair(100000, 51, 52, sub_rate=0.07, mask_frac=0.1)
    c = Case(t, q, chunk=50000).oracle_setup(oracle).engine_setup(clean)
    seen = set()
    hits = [c.oracle_saf(c.host_seeds(s, e, rev), rev)[1]["num_hits"] for rev in (False, True) for (s, e) in c.chunks()]
    assert min(hits) > 1000
    # far above eight iterations for every chunk / three to five / two or three / no split at all
    for mh in (min(hits) // 10, max(hits) // 3, max(hits) - 1, 1 << 30):
        c.E.set_max_hits(mh)
        for rev in (False, True):
            for (s, e) in c.chunks():
                want, st = c.oracle_saf(c.host_seeds(s, e, rev), rev, max_hits=mh)
                got = c.E.SeedAndFilterRange(s, e, rev, 0)
                es = c.E.last_call_stats()
                assert seg_equal(got, want), (mode, mh, rev, s, e)
                assert es["num_iter"] == st["num_iter"], (mode, mh, rev, s, e, es["num_iter"], st["num_iter"])
                general = bool(es["path_flags"] & c.E.PATH_GENERAL_FALLBACK) or es["lookup_path"] == 0
                if mode and st["num_hits"] >= mh:
                    # planned table-direct up to eight iterations, by the general path above
                    assert general == (st["num_hits"] // mh + 2 > 8), (mode, mh, rev, s, e, st["num_hits"], es)
                    seen.add(general)
                elif mode:
                    assert not general
    assert not mode or seen == {False, True}, seen   # both sides of the limit were reached
    c.E.set_max_hits(0)


@pytest.mark.parametrize("env", [{"SEGALIGN_AMD_SPEC_RECS": "8"},        # most records travel in the second copy
                                 {"SEGALIGN_AMD_DEDUP_SEG_MAX": "4"},     # a segment too large for the LDS chain: library sorts
                                 {"SEGALIGN_AMD_SPEC_DEDUP": "0"},        # chain launched after the survivor count is known
                                 {"SEGALIGN_AMD_NO_SMALL_DEDUP": "1"},    # library sorts only
                                 {"SEGALIGN_AMD_L2_CAP": "1024"}])        # 4 records per sub-list: overflow, regrow, rerun
def test_every_output_path_of_a_multi_chunk_call(oracle, clean, env):
    """The ways the survivors of a call can reach the host -- speculative LDS chain with one or two copies, the same chain after
    a sync, the library-sort fallback -- must all give the oracle's vectors."""
    with_env(env)
    t, q = synth.make_pair(200000, 71, 72, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=500)
    c = Case(t, q, chunk=12000).oracle_setup(oracle).engine_setup(clean)
    ch = c.chunks()
    assert len(ch) == 17
    total = 0
    for rev in (False, True):
        wants = [c.oracle_saf(c.host_seeds(s, e, rev), rev)[0] for (s, e) in ch]
        outs = c.E.SeedAndFilterChunks(ch[0][0], ch[15][1], rev, 0)
        for j in range(16):
            assert seg_equal(outs[j], wants[j]), (env, rev, j)
            total += outs[j].size - 1
        assert seg_equal(c.E.SeedAndFilterRange(ch[16][0], ch[16][1], rev, 0), wants[16]), (env, rev)
    assert total > 64


def test_multi_chunk_call_with_empty_chunks_and_a_max_hits_split(oracle, clean):
    """One 16-chunk call whose chunks are not alike: chunks without a single seed (runs of N: NULL / 0 in their slots, no header),
    a soft-masked chunk, and -- second pass -- a MAX_HITS low enough that one chunk needs more than two reference iterations,
    which sends the WHOLE call down the general path (per-seed-word prefixes, batches of <= 8 iterations).  Every slot must equal
    the single-chunk call and the oracle."""
    with_env({})
    t, q = synth.make_pair(300000, 61, 62, sub_rate=0.08, mask_frac=0.1, records=2, indel_every=600)
    q = q.copy()
    chunk = 20000
    q[3 * chunk:5 * chunk] = ord("N")                      # chunks 3, 4: no seeds at all
    q[9 * chunk:10 * chunk] = np.frombuffer(bytes(q[9 * chunk:10 * chunk]).lower(), dtype=np.uint8)  # chunk 9: soft-masked
    c = Case(t, q, chunk=chunk).oracle_setup(oracle).engine_setup(clean)
    E = c.E
    ch = c.chunks()
    assert len(ch) >= 15
    for mh in (1 << 30, 6000):
        E.set_max_hits(mh)
        for rev in (False, True):
            wants = [c.oracle_saf(c.host_seeds(s, e, rev), rev, max_hits=mh)[0] if c.host_seeds(s, e, rev).size else None for (s, e) in ch]
            k = min(len(ch), 16)
            outs = E.SeedAndFilterChunks(ch[0][0], ch[k - 1][1], rev, 0)
            n_empty = 0
            for j in range(k):
                if wants[j] is None:
                    assert outs[j].size == 0, (mh, rev, j)
                    n_empty += 1
                else:
                    assert seg_equal(outs[j], wants[j]), (mh, rev, j)
                    assert seg_equal(E.SeedAndFilterRange(ch[j][0], ch[j][1], rev, 0), wants[j]), (mh, rev, j)
            assert n_empty >= 2 or rev  # (the N run lies elsewhere on the minus strand's chunk grid)
    E.set_max_hits(0)


@pytest.mark.parametrize("t_run,q_run,min_hits", [(3000, 400, 1_000_000),   # > 2048 survivors in one dedup segment: library sorts
                                                  (1300, 200, 200_000)])    # 1024 .. 2048: the LDS chain's second tile (in-place unique)
@pytest.mark.parametrize("mode,env", MODES)
def test_low_complexity_mega_buckets(oracle, clean, mode, env, t_run, q_run, min_hits):
    """Unmasked low-complexity sequence: a 3 kb poly-A run and a 1.5 kb (CT)n microsatellite in the target put thousands of
    positions into single buckets; the query's own runs then produce ~10^6 hits from a few hundred positions, thousands of
    diagonals whose candidates all extend to overlapping HSPs (chain shortcut, > 2048 survivors in one dedup segment ->
    library-sort fallback), a record of the probe that spans many 64-hit buffers, and scores far above 3*hspthresh."""
    with_env(env)
    rng = np.random.default_rng(91)
    t, q = synth.make_pair(120000, 91, 92, sub_rate=0.08, mask_frac=0.05, records=2, indel_every=700)
    t, q = t.copy(), q.copy()
    t[20000:20000 + t_run] = ord("A")
    t[70000:71500] = np.tile(np.frombuffer(b"CT", dtype=np.uint8), 750)
    q[50000:50000 + q_run] = ord("A")
    q[50000 + q_run // 2 - 50] = ord("G")                 # one transition inside the run
    q[90000:90300] = np.tile(np.frombuffer(b"CT", dtype=np.uint8), 150)
    c = Case(t, q, chunk=30000).oracle_setup(oracle).engine_setup(clean)
    E = c.E
    assert E.lookup_mode() == mode
    hits = 0
    for rev in (False, True):
        wants = []
        for (s, e) in c.chunks():
            want, st = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), want), (mode, rev, s, e)
            assert E.last_call_stats()["num_hits"] == st["num_hits"]
            hits = max(hits, st["num_hits"])
            wants.append(want)
        ch = c.chunks()
        outs = E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)
        for j, w in enumerate(wants):
            assert seg_equal(outs[j], w), (mode, rev, j)
    assert hits > min_hits


@pytest.mark.parametrize("mode,env", MODES)
def test_lookup_paths_repeat_masker(oracle, clean, mode, env):
    with_env(env)
    unit = synth.random_dna(600, 77)
    t = synth.random_dna(200000, 15)
    rng = np.random.default_rng(3)
    for i in range(90):
        p = int(rng.integers(0, t.size - 700))
        cp = synth.mutate(unit, 500 + i, 0.05)
        t[p:p + cp.size] = cp if i % 3 else synth.reverse_complement(cp)
    t = synth.soft_mask(t, 5, 0.04)
    c = Case(t, t, chunk=30000).oracle_setup(oracle).engine_setup(clean)
    c.E.RmSendQueryWriteRequest()
    assert c.E.lookup_mode() == mode
    L = t.size
    for (s, e, ws, we, strands) in ((0, L - 19, 0, L, 3), (40000, 130000, 20000, 150000, 3), (100000, 160000, 0, 90000, 1),
                                    (0, 70000, 120000, L, 2)):
        want, wt = model_mask_interval(c, oracle, s, e, ws, we, strands, 1)
        got, gt = c.E.RmMaskInterval(s, e, ws, we, strands, 1)
        assert as_list(got) == as_list(want) and gt == wt, (mode, s, e, ws, we, strands)
    c.E.RmClearQuery()


def test_query_block_beyond_the_2bit_copy_limit_takes_the_general_path(oracle, clean):
    """The class filter addresses the sixteen 2-bit copies of a query strand with one 32-bit offset (4 GiB: blocks of ~1 Gbp).  A
    block beyond that must not take the table-direct path against the context table (its run entries carry no plain positions) --
    it falls back to the reference-shaped path, same results.  The limit is lowered through the test option q2_limit_mb."""
    with_env({})
    E = clean
    t, q = synth.make_pair(300000, 61, 62, sub_rate=0.09, mask_frac=0.1, records=2, indel_every=500)
    try:
        E.set_option("cls_one_copy", 2)  # the sixteen shifted copies (the default one-copy form has no such limit)
        E.set_option("q2_limit_mb", 1)  # 16 copies x ~75 KB = 1.2 MB > 1 MB
        c = Case(t, q, chunk=100000).oracle_setup(oracle).engine_setup(E)
        assert E.lookup_mode() == 2     # the table itself is the context table ...
        n = 0
        for rev in (False, True):
            for (s, e) in c.chunks():
                seeds = c.host_seeds(s, e, rev)
                want, _ = c.oracle_saf(seeds, rev)
                assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), want)
                assert E.last_call_stats()["lookup_path"] == 0   # ... but the calls cannot use it
                assert seg_equal(E.SeedAndFilter(seeds, rev, 0), want)
                assert E.last_call_stats()["lookup_path"] == 0
                n += want.size - 1
            ch = c.chunks()
            outs = E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)
            for j, (s, e) in enumerate(ch):
                assert seg_equal(outs[j], c.oracle_saf(c.host_seeds(s, e, rev), rev)[0])
        assert n > 50
    finally:
        E.reset_option("q2_limit_mb")
        E.reset_option("cls_one_copy")


@pytest.mark.parametrize("mode,env", MODES)
def test_passes_that_cannot_hold_their_chunks_halve_themselves(oracle, clean, mode, env):
    """A multi-chunk pass with chunks far above MAX_HITS (more than eight reference iterations: those need the general path's plan, hazard H4), and a pass of
    more than 32 chunks on the general path, are cut in halves until every piece fits (api_calls.hip chunks_pass): every chunk's
    vector must still be what a call of its own returns, the statistics the sums."""
    with_env(env)
    t, q = synth.make_pair(260000, 81, 82, sub_rate=0.07, mask_frac=0.1, records=2, indel_every=600)
    c = Case(t, q, chunk=5000).oracle_setup(oracle).engine_setup(clean)  # 52 chunks
    E = c.E
    ch = c.chunks()
    assert len(ch) > 40
    try:
        for mh in (300, 2500, 1 << 30):   # ~half of the chunks hold more hits than 2500 (three iterations: they stay table-direct); every chunk far more than 300
            E.set_max_hits(mh)
            for rev in (False, True):
                wants, hits = [], 0
                for (s, e) in ch[:48]:
                    w, st = c.oracle_saf(c.host_seeds(s, e, rev), rev, max_hits=mh)
                    wants.append(w)
                    hits += st["num_hits"]
                outs = E.SeedAndFilterChunks(ch[0][0], ch[47][1], rev, 0)   # 48 chunks in one entry call
                st = E.last_call_stats()
                for j, w in enumerate(wants):
                    assert seg_equal(outs[j], w), (mode, mh, rev, j)
                assert st["num_hits"] == hits
                if mh == 300 and mode:
                    assert st["path_flags"] & E.PATH_GENERAL_FALLBACK   # chunks that need more than eight iterations went down the general path ...
                if mh == 2500 and mode:
                    assert not (st["path_flags"] & E.PATH_GENERAL_FALLBACK) and st["num_iter"] > 2 * 48   # ... three or four are planned table-direct
    finally:
        E.set_max_hits(0)
