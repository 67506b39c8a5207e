"""GPU: the X-drop FILTER levels never reject a hit the reference would keep.

Final-output parity cannot see a wrongly rejected hit whenever a sibling hit of the same HSP survives (every hit inside an HSP
extends to the same record), so the filters are audited directly: with option audit_cap the engine records every hit its
filter levels (class filter on the 28-byte context records, extend.hip 1d; packed second level, 1b; or the 32-byte pair-scoring
form, 1c) REJECT, and each of them is extended here by the oracle's scalar find_hsps (src/seed_filter.cu:232-652) -- none
may pass.  Hits near soft-masked runs, N runs, record separators and the block edges are all in the sample."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def audited(engine):
    engine.set_option("audit_cap", 1 << 22)
    yield engine
    engine.ShutdownProcessor()
    engine.reset_option(None)


def audit_case(O, E, c, chunks_per_call=1, strands=(False, True)):
    rejected = 0
    ch = c.chunks()
    for rev in strands:
        qcodes = c.o_qrc if rev else c.o_q
        for g in range(0, len(ch), chunks_per_call):
            grp = ch[g:g + chunks_per_call]
            wants = [c.oracle_saf(c.host_seeds(s, e, rev), rev)[0] for (s, e) in grp]
            if chunks_per_call == 1:
                outs = [E.SeedAndFilterRange(grp[0][0], grp[0][1], rev, 0)]
            else:
                outs = E.SeedAndFilterChunks(grp[0][0], grp[-1][1], rev, 0)
            for got, want in zip(outs, wants):
                assert seg_equal(got, want)
            if int(E.last_call_stats()["num_hits"]) == 0:
                continue  # (a call without hits never starts the filter: the audit list still holds the previous call's)
            pairs, n = E.get_audit()
            assert n == pairs.shape[0], "audit list overflowed: raise audit_cap"
            assert n <= int(E.last_call_stats()["num_hits"])
            ok, recs = O.extend_hits_pass(c.o_ref, qcodes, c.sub_mat, pairs, xdrop=c.xdrop, hspthresh=c.hspthresh, noentropy=c.noentropy)
            bad = np.nonzero(ok)[0]
            assert bad.size == 0, (rev, grp[0], pairs[bad[:5]], recs[bad[:5]])
            rejected += n
    return rejected


# (l2_right_state = 1: the second level resumes an open RIGHT walk behind the class filter's 54 context bases from its packed state --
#  off by default since it moves no clock, profiles/r06/ab_l2state_prio.txt -- audited here like the default form)
@pytest.mark.parametrize("env_opt", [{}, {"no_ctx": 1}, {"l2_right_state": 1}])
@pytest.mark.parametrize("noentropy", [False, True])
def test_filters_reject_nothing_that_passes(oracle, audited, env_opt, noentropy):
    for k, v in env_opt.items():
        audited.set_option(k, v)
    t, q = synth.make_pair(300000, 81, 82, sub_rate=0.13, mask_frac=0.2, records=3, indel_every=120, n_runs=3)
    c = Case(t, q, chunk=60000, noentropy=noentropy).oracle_setup(oracle).engine_setup(audited)
    assert audited.lookup_mode() == (1 if env_opt.get("no_ctx") else 2)
    n = audit_case(oracle, audited, c)
    assert n > 50000  # the bulk of the ~140 k hits is rejected by the filters, and all of them were checked


@pytest.mark.parametrize("in_query", [False, True])
def test_filters_on_a_multi_chunk_call_with_iupac_and_low_thresholds(oracle, audited, in_query):
    """Other IUPAC letters (code X scores -100 against ACGT: the class bound has to cover it), a low hspthresh and xdrop (many
    near-threshold hits), sixteen chunks per call.  With such letters in the QUERY only the plus strand is comparable: the
    reference's host RevComp shifts the minus-strand arena there (hazard H14)."""
    t, q = synth.make_pair(200000, 91, 92, sub_rate=0.10, mask_frac=0.1, records=2, indel_every=200)
    q = q.copy()
    t = t.copy()
    rng = np.random.default_rng(5)
    for arr in ((t, q) if in_query else (t,)):
        pos = rng.integers(0, arr.size, 400)
        arr[pos] = np.frombuffer(b"RYKMSW", dtype=np.uint8)[rng.integers(0, 6, 400)]
    c = Case(t, q, chunk=8000, xdrop=500, hspthresh=2200).oracle_setup(oracle).engine_setup(audited)
    n = audit_case(oracle, audited, c, chunks_per_call=16, strands=(False,) if in_query else (False, True))
    assert n > 5000


def test_both_context_layouts_reject_nothing_that_passes(oracle, audited):
    """Round 5's context records hold the 64 bases in FRONT of the seed window, which the class filter bounds by seed_size x the
    largest class score instead of walking it (kernels.h CtxRec); option ctx_skip_seed = 0 is round 4's layout (the 64 bases left of
    the anchor, seed window included).  Same vectors, every rejected hit audited under both -- near the block start too, where the
    longer reach of the new left context runs into the pad --, with other seed sizes (14of22: a 22-base window), and the new layout
    never forwards more hits to the second level (on a 100 Mbp target, where chance hits dominate: half as many, profiles/r05)."""
    t, q = synth.make_pair(240000, 91, 92, sub_rate=0.10, mask_frac=0.15, records=2, indel_every=200, n_runs=2)
    q = q.copy()
    q[:400] = t[:400]            # homology that touches position 0 of both blocks: left walks end at the block edge
    fwd = {}
    for skip in (1, 0):
        audited.ShutdownProcessor()
        audited.set_option("ctx_skip_seed", skip)
        c = Case(t, q, chunk=60000).oracle_setup(oracle).engine_setup(audited)
        assert audited.lookup_mode() == 2
        n = audit_case(oracle, audited, c, chunks_per_call=2)
        assert n > 50000
        tot_f = tot_h = 0
        for rev in (False, True):
            for (s, e) in c.chunks():
                audited.SeedAndFilterRange(s, e, rev, 0)
                st = audited.last_call_stats()
                tot_f += st["num_forwarded"]
                tot_h += st["num_hits"]
        fwd[skip] = tot_f / max(tot_h, 1)
    assert fwd[1] <= fwd[0], fwd  # (a 240 kbp target: most hits are homologous and forwarded under either layout)
    # another seed window: 14of22 (src/main.cpp:164-167), transitions on
    import re
    src = open(__file__.replace("test_gpu_filter_audit.py", "test_gpu_edge_cases.py")).read()
    shape22 = re.search(r'"([T0]{22})"', src).group(1)
    audited.ShutdownProcessor()
    audited.set_option("ctx_skip_seed", 1)
    t2, q2 = synth.make_pair(120000, 93, 94, sub_rate=0.08, mask_frac=0.1, indel_every=300)
    c = Case(t2, q2, chunk=60000, shape=shape22).oracle_setup(oracle).engine_setup(audited)
    if audited.lookup_mode() == 2:
        assert audit_case(oracle, audited, c) > 1000
