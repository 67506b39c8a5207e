"""GPU: a C++ host that uses ONLY the reference's own symbols (g_* function pointers + GenerateSeedPosTable from
include/segalign_amd_compat.hpp), with several host threads like the TBB seeder bodies, gives HSP lists identical
to the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import Case, SHAPE_12OF19
from segalign_amd import synth
from test_compat_header import build_driver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads", [1, 4])
def test_reference_symbols_end_to_end(oracle, tmp_path, threads):
    t, q = synth.make_pair(240000, 31, 32, sub_rate=0.10, mask_frac=0.1, records=2, indel_every=500)
    chunk = 60000
    c = Case(t, q, chunk=chunk).oracle_setup(oracle)
    tp, qp = tmp_path / "t.txt", tmp_path / "q.txt"
    tp.write_bytes(t.tobytes())
    qp.write_bytes(q.tobytes())
    exe = build_driver()
    out = subprocess.check_output([exe, str(tp), str(qp), SHAPE_12OF19, str(chunk), "1", str(threads)],
                                  stderr=subprocess.DEVNULL).decode().split("\n")
    i = 0
    total = 0
    while i < len(out) and out[i]:
        tag, rev, s, e, n, hits, path = out[i].split()
        assert tag == "C"
        rev, s, e, n, hits = int(rev), int(s), int(e), int(n), int(hits)
        got = np.array([tuple(int(x) for x in out[i + 1 + j].split()) for j in range(n)],
                       dtype=[("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4")])
        seeds = c.host_seeds(s, e, bool(rev))
        want, st = c.oracle_saf(seeds, bool(rev))
        assert hits == st["num_hits"]
        # the reference's own hot symbol gets the fast path: a host seed vector that is what seeder.cpp emits is looked up
        # table-direct with target context (sa_call_stats.lookup_path), not through the seed-word path
        assert seeds.size == 0 or int(path) == 2, (rev, s, e, path)
        assert got.size == want.size - 1 and np.all(got == want[1:]), (rev, s, e)
        total += n
        i += 1 + n
    assert total > 0


@pytest.mark.parametrize("threads", [1, 3])
def test_repeat_masker_reference_symbols_end_to_end(oracle, tmp_path, threads):
    """The same host built against the REPEAT MASKER's symbols (g_SendQueryWriteRequest(), g_SeedAndFilter(seeds, rev,
    ref_start, ref_end), g_ClearQuery(); repeat_masker_src/seed_filter.h:4-14) and driven like its seeder
    (repeat_masker_src/seeder.cpp:73-146): every windowed call equals the oracle, 64-bit header included."""
    O = oracle
    unit = synth.random_dna(500, 77)
    t = synth.random_dna(150000, 15)
    rng = np.random.default_rng(3)
    for i in range(70):  # a diverged repeat family in both orientations
        p = int(rng.integers(0, t.size - 600))
        cp = synth.mutate(unit, 500 + i, 0.06)
        t[p:p + cp.size] = cp if i % 3 else synth.reverse_complement(cp)
    t = synth.soft_mask(t, 5, 0.05)
    chunk, ws, we = 40000, 20000, 140000
    c = Case(t, t, chunk=chunk).oracle_setup(oracle)
    rc_ascii = np.frombuffer(O.rev_comp_ascii(t.tobytes(), 0, t.size), dtype=np.uint8)
    o_rc = O.rev_comp_codes(c.o_ref)
    tp = tmp_path / "t.txt"
    tp.write_bytes(t.tobytes())
    exe = build_driver(rm=True)
    out = subprocess.check_output([exe, str(tp), str(tp), SHAPE_12OF19, str(chunk), "1", str(threads), str(ws), str(we)],
                                  stderr=subprocess.DEVNULL).decode().split("\n")
    i = total = ncalls = 0
    while i < len(out) and out[i]:
        tag, rev, s, e, n, hits, path = out[i].split()
        assert tag == "C"
        rev, s, e, n, hits = int(rev), int(s), int(e), int(n), int(hits)
        assert int(path) == 2, (rev, s, e, path)  # (table-direct with target context for the repeat masker's g_SeedAndFilter too)
        got = np.array([tuple(int(x) for x in out[i + 1 + j].split()) for j in range(n)], dtype=O.SEG_DTYPE)
        buf = rc_ascii if rev else t
        seeds = O.make_seeds(buf.tobytes(), 0, s, e, 19, c.kmer_size, True)
        want, st = O.seed_and_filter(c.o_ref, o_rc if rev else c.o_ref, c.o_index, c.o_pos, seeds, c.sub_mat, rm=(bool(rev), ws, we))
        assert hits == st["num_hits"], (rev, s, e)
        assert got.size == want.size - 1 and np.all(got == want[1:]), (rev, s, e)
        total += n
        ncalls += 1
        i += 1 + n
    assert ncalls == 2 * len(c.chunks()) and total > 20
