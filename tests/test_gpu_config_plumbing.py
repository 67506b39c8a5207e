"""GPU, BASELINE.json configs[0] (the reference's own CPU-runnable plumbing case, SURVEY 8d "Config 1"): 1 Mbp uniform
ACGT target (PRNG seed 1) x its copy with 15 % substitutions and one 1-10 bp indel about every 500 bp (seed 2); 12of19,
HOXD70 defaults.  Small enough to be EXHAUSTIVE: every 250 kbp chunk of both strands, HIP == oracle, through every entry
point, and the interval entry equals the concatenation of the chunk calls (src/seeder.cpp:80-85,115-120)."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plumbing(oracle, engine):
    t, q = synth.make_pair(1_000_000, 1, 2, sub_rate=0.15, indel_every=500, invert_frac=0.0)
    c = Case(t, q, chunk=250000).oracle_setup(oracle).engine_setup(engine)
    yield c
    engine.ShutdownProcessor()


def test_configs0_tables_and_encoding(plumbing):
    c = plumbing
    assert np.array_equal(c.E.copy_ref_codes(), c.o_ref)
    assert np.array_equal(c.E.copy_query_codes(0, False), c.o_q) and np.array_equal(c.E.copy_query_codes(0, True), c.o_qrc)
    assert np.array_equal(c.E.copy_index_table(), c.o_index) and np.array_equal(c.E.copy_pos_table(), c.o_pos)


def test_configs0_whole_workload_bit_exact(plumbing):
    c, E = plumbing, plumbing.E
    q_len = c.query.size - c.seed_size
    per_strand = {False: [], True: []}
    hits = 0
    for rev in (False, True):
        for (s, e) in c.chunks():
            seeds = c.host_seeds(s, e, rev)
            want, st = c.oracle_saf(seeds, rev)
            assert np.array_equal(E.device_make_seeds(s, e, rev, 0), seeds)
            assert seg_equal(E.SeedAndFilter(seeds, rev, 0), want), (rev, s, e)
            assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), want), (rev, s, e)
            per_strand[rev].append(want[1:])
            hits += st["num_hits"]
        outs = E.SeedAndFilterChunks(0, q_len, rev, 0)
        for got, (s, e) in zip(outs, c.chunks()):
            want, _ = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            assert seg_equal(got, want)
    # the reference seeder walks the minus strand in rc interval coordinates (seeder.cpp:33-34): with ONE interval covering
    # [0, q_len) both strands see the same chunk bounds
    fw, rc, st = E.SeedInterval(0, q_len, q_len, E.STRAND_BOTH, 0, 2)
    assert np.array_equal(fw, np.concatenate(per_strand[False])) and np.array_equal(rc, np.concatenate(per_strand[True]))
    assert st["num_hits"] == hits
    assert fw.size > 300  # 15 % divergence, an indel every ~500 bp: hundreds of HSPs on the plus strand
