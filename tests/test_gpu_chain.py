"""GPU: the chain shortcut of the exact stage (extend.hip 2b) on inputs built to stress its premises -- mismatch bursts
whose drop sits right at the X-drop threshold (HSP ends depend on the anchor), candidates crowding one diagonal,
indels that move the diagonal every few dozen bases -- with the shortcut on (default) and off."""
import os

import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def borderline_pair(seed, n=60000):
    """Copy with bursts of 6-10 consecutive substitutions every 40-90 bases: each burst costs ~650-1250 points, i.e.
    sometimes below and sometimes above xdrop = 910, so whether a walk crosses a burst depends on where it started."""
    rng = np.random.default_rng(seed)
    t = synth.random_dna(n, seed)
    q = t.copy()
    p = 50
    while p < n - 20:
        k = int(rng.integers(6, 11))
        idx = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), q[p:p + k])
        q[p:p + k] = np.frombuffer(b"ACGT", dtype=np.uint8)[(idx + rng.integers(1, 4, size=k)) % 4]
        p += int(rng.integers(40, 90))
    return t, q


def run(engine, oracle, t, q, chain, **kw):
    if chain:
        os.environ.pop("SEGALIGN_AMD_NO_CHAIN", None)
    else:
        os.environ["SEGALIGN_AMD_NO_CHAIN"] = "1"
    c = Case(t, q, **kw).oracle_setup(oracle).engine_setup(engine)
    try:
        outs, surv = [], 0
        run.flags = 0
        for rev in (False, True):
            for (s, e) in c.chunks():
                seeds = c.host_seeds(s, e, rev)
                if seeds.size == 0:
                    continue
                want, st = c.oracle_saf(seeds, rev)
                got = c.E.SeedAndFilter(seeds, rev, 0)
                assert seg_equal(got, want), (chain, rev, s, e)
                surv += c.E.last_call_stats()["num_survivors"]
                run.flags |= c.E.last_call_stats()["path_flags"]
                outs.append(got)
        return outs, surv
    finally:
        engine.ShutdownProcessor()
        os.environ.pop("SEGALIGN_AMD_NO_CHAIN", None)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_borderline_xdrop_bursts(oracle, engine, seed):
    t, q = borderline_pair(100 + seed)
    on, s_on = run(engine, oracle, t, q, True, chunk=30000)
    off, s_off = run(engine, oracle, t, q, False, chunk=30000)
    assert all(np.array_equal(a, b) for a, b in zip(on, off))
    assert 0 < s_on <= s_off  # the shortcut only ever removes exact duplicates before the dedup stage


@pytest.mark.parametrize("xdrop,hspthresh", [(300, 2000), (910, 3000), (2000, 3000)])
def test_dense_single_diagonal_and_shifting_diagonals(oracle, engine, xdrop, hspthresh):
    # one 40 kb collinear diagonal (3 % substitutions) ...
    t1, q1 = synth.make_pair(40000, 7, 8, sub_rate=0.03, invert_frac=0.0)
    on, s_on = run(engine, oracle, t1, q1, True, chunk=20000, xdrop=xdrop, hspthresh=hspthresh)
    off, s_off = run(engine, oracle, t1, q1, False, chunk=20000, xdrop=xdrop, hspthresh=hspthresh)
    assert all(np.array_equal(a, b) for a, b in zip(on, off)) and s_on < s_off
    # ... and indels every ~40 bases: the diagonal moves all the time, groups are tiny
    t2, q2 = synth.make_pair(40000, 9, 10, sub_rate=0.02, indel_every=40, invert_frac=0.3, invert_block=4000)
    on, _ = run(engine, oracle, t2, q2, True, chunk=20000, xdrop=xdrop, hspthresh=hspthresh)
    off, _ = run(engine, oracle, t2, q2, False, chunk=20000, xdrop=xdrop, hspthresh=hspthresh)
    assert all(np.array_equal(a, b) for a, b in zip(on, off))


def test_chain_with_iteration_split(oracle, engine):
    """Candidates of different reference iterations (MAX_HITS split) must never share a chain."""
    t, q = synth.make_pair(50000, 11, 12, sub_rate=0.04, invert_frac=0.0)
    engine.set_max_hits(3000)
    try:
        c = Case(t, q, chunk=25000).oracle_setup(oracle).engine_setup(engine)
        for (s, e) in c.chunks():
            seeds = c.host_seeds(s, e, False)
            want, st = c.oracle_saf(seeds, False, max_hits=3000)
            assert st["num_iter"] > 3
            assert seg_equal(c.E.SeedAndFilter(seeds, False, 0), want)
    finally:
        engine.ShutdownProcessor()
        engine.set_max_hits(0)


def test_candidate_list_larger_than_the_chain_buffers(oracle, engine):
    """more candidates than SEGALIGN_AMD_CHAIN_CAP: the chain kernels leave the batch alone at first (the count lives on the device)
    and the host runs the chain stages over the candidate list slice by slice (core.hip), same output"""
    t = synth.random_dna(50000, 31)
    q = synth.mutate(t, 32, 0.03)
    os.environ["SEGALIGN_AMD_CHAIN_CAP"] = "500"
    try:
        outs, surv = run(engine, oracle, t, q, True)
    finally:
        os.environ.pop("SEGALIGN_AMD_CHAIN_CAP", None)
    assert surv > 0 and run.flags & engine.PATH_CHAIN_SLICED


def test_crowded_buckets_keep_the_windows_of_one_diagonal_apart(oracle, engine):
    """A collinear pair puts every candidate on ONE diagonal; with few hash buckets (option chain_buckets) a bucket holds several
    512-position windows of it.  The sort key must keep the windows apart -- ordered by (diagonal, position mod 512) alone they
    interleave, every link test fails and every candidate is extended (the first cut of the 32-bit keys did exactly that: same
    output, 8 x the extensions).  Same output with 128 buckets as with the default, and nearly as few extensions."""
    t1, q1 = synth.make_pair(60000, 41, 42, sub_rate=0.03, invert_frac=0.0)
    base, s_base = run(engine, oracle, t1, q1, True, chunk=30000)
    os.environ["SEGALIGN_AMD_CHAIN_BUCKETS"] = "128"   # ~60 windows of 512 candidates per call in 128 buckets: a dozen shared buckets,
    os.environ["SEGALIGN_AMD_CHAIN_GROUP_MAX"] = "4096"  # each still within what a workgroup sorts (larger ones are left unsorted)
    try:
        few, s_few = run(engine, oracle, t1, q1, True, chunk=30000)
    finally:
        os.environ.pop("SEGALIGN_AMD_CHAIN_BUCKETS", None)
        os.environ.pop("SEGALIGN_AMD_CHAIN_GROUP_MAX", None)
    off, s_off = run(engine, oracle, t1, q1, False, chunk=30000)
    assert all(np.array_equal(a, b) for a, b in zip(base, few))
    assert s_base * 5 < s_off          # the shortcut is worth something on this input at all ...
    assert s_few <= 1.25 * s_base + 64  # ... and crowding the buckets costs next to nothing (interleaved windows: thousands)


@pytest.mark.parametrize("threads", [64, 128, 512])
def test_chain_sort_and_link_with_any_workgroup_size(oracle, engine, threads):
    """Option chain_sort_threads (64..512): the fused sort + link kernel stages its 128-entry score table with a strided loop, so a
    64-thread workgroup fills the terminator half too (round 4's advisor: with `threadIdx.x < 128` it stayed uninitialised, link
    tests misfired and run members were dropped), and chain_group_max = 4096 announces its 78 KB of dynamic LDS.  Same vectors as
    the default, the oracle's, and about as few extensions."""
    t1, q1 = synth.make_pair(60000, 41, 42, sub_rate=0.03, invert_frac=0.0)
    base, s_base = run(engine, oracle, t1, q1, True, chunk=30000)
    os.environ["SEGALIGN_AMD_CHAIN_SORT_THREADS"] = str(threads)
    os.environ["SEGALIGN_AMD_CHAIN_GROUP_MAX"] = "4096"
    os.environ["SEGALIGN_AMD_CHAIN_BUCKETS"] = "128"
    try:
        got, s_got = run(engine, oracle, t1, q1, True, chunk=30000)   # (run() holds every vector against the oracle)
    finally:
        for k in ("SEGALIGN_AMD_CHAIN_SORT_THREADS", "SEGALIGN_AMD_CHAIN_GROUP_MAX", "SEGALIGN_AMD_CHAIN_BUCKETS"):
            os.environ.pop(k, None)
    assert all(np.array_equal(a, b) for a, b in zip(base, got))
    assert s_got <= 1.25 * s_base + 64
