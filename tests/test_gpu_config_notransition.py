"""GPU, BASELINE.json configs[4]: the ce11 x cb4 stand-in with --notransition --step=1 (one seed word per query position,
src/seeder.cpp:57-61 without the :62-70 neighbour loop; HOXD70 defaults).  Same shape as test_gpu_fullsize.py: the oracle
cannot run the whole workload in seconds, so full-size parity = size-independent properties + bit-exact agreement on
whole chunk calls of both strands (table copied from the device)."""
import numpy as np
import pytest

from segalign_amd import shard

pytestmark = pytest.mark.gpu

SHAPE = "TTT0T00TT00T0T0TTTT"


@pytest.fixture(scope="module")
def notrans(oracle, engine, standin_100mbp):
    E, O = engine, oracle
    target, query = standin_100mbp
    sub_mat = O.build_sub_mat(910)
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    O.generate_shape_pos(SHAPE)
    E.InitializeProcessor(False, 250000, 19, sub_mat, 910, 3000, False)  # --notransition
    keep = E.SendRefWriteRequest(target, 0, target.size)
    E.GenerateSeedPosTable(keep, 0, target.size, 1, 19, k)               # --step=1
    E.SendQueryWriteRequest(query, 0, query.size, 0)
    yield dict(E=E, O=O, target=target, query=query, sub_mat=sub_mat, k=k, index=E.copy_index_table(), pos=E.copy_pos_table(),
               rcodes=E.copy_ref_codes(),
               rc_ascii=np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8))
    E.ShutdownProcessor()


def test_configs4_seed_words_one_per_position(notrans):
    E, O, query = notrans["E"], notrans["O"], notrans["query"]
    for rev, buf in ((False, query), (True, notrans["rc_ascii"])):
        a, b = 12_345_678, 12_595_678
        dev = E.device_make_seeds(a, b, rev, 0, per=1)
        host = O.make_seeds(buf.tobytes(), 0, a, b, 19, notrans["k"], False)
        assert host.size <= b - a and np.array_equal(dev, host)
        assert np.all(np.diff((host & np.uint64(0xFFFFFFFF)).astype(np.int64)) > 0)  # one word per position, ascending


@pytest.mark.parametrize("rev", [False, True])
def test_configs4_whole_interval_chunks_bit_exact_vs_oracle(notrans, rev):
    """Every 250 kbp chunk of a 2 Mbp stretch (8 calls per strand): drop-in entry (host seed words), device-seeded entry
    and the multi-chunk entry against the oracle."""
    E, O, query = notrans["E"], notrans["O"], notrans["query"]
    qlen = query.size - 19
    qcodes = E.copy_query_codes(0, rev)
    buf = notrans["rc_ascii"] if rev else query
    chunks = shard.chunks_of((60_000_000, 62_000_000), 250000, qlen, rev)
    wants, hits, hsps = [], 0, 0
    for (a, b) in chunks:
        seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, notrans["k"], False)
        want, st = O.seed_and_filter(notrans["rcodes"], qcodes, notrans["index"], notrans["pos"], seeds, notrans["sub_mat"])
        got = E.SeedAndFilter(seeds, rev, 0)
        assert got.shape == want.shape and np.all(got == want), (a, b)
        got2 = E.SeedAndFilterRange(a, b, rev, 0)
        assert got2.shape == want.shape and np.all(got2 == want), (a, b)
        wants.append(want)
        hits += st["num_hits"]
        hsps += want.size - 1
    assert hits > 4_000_000 and hsps > 50
    assert len(chunks) == 8
    for k in (4, 8):  # four / all eight chunks of the strand in one pass over the kernels
        for g in range(0, len(chunks), k):
            outs = E.SeedAndFilterChunks(chunks[g][0], chunks[min(g + k - 1, len(chunks) - 1)][1], rev, 0)
            for j, w in enumerate(wants[g:g + k]):
                assert outs[j].shape == w.shape and np.all(outs[j] == w), (k, g, j)


def test_configs4_interval_properties(notrans):
    """One whole 10 Mbp interval on both strands (the bench step of this workload): counts add up, every HSP passes the
    threshold and rescoring the raw interval reproduces scores above the entropy band."""
    E = notrans["E"]
    M = notrans["sub_mat"].reshape(8, 8)
    qlen = notrans["query"].size - 19
    fw, rc, st = E.SeedInterval(20_000_000, 30_000_000, qlen, E.STRAND_BOTH, 0, 2)
    fw2, rc2, st2 = E.SeedInterval(20_000_000, 30_000_000, qlen, E.STRAND_BOTH, 0, 1)
    assert np.array_equal(fw, fw2) and np.array_equal(rc, rc2) and st["num_hits"] == st2["num_hits"]
    assert st["num_seeds"] <= 2 * 10_000_000 and st["num_seeds"] > 10_000_000  # one word per valid position per strand
    assert fw.size > 500 and rc.size > 100
    for rev, body in ((False, fw), (True, rc)):
        qcodes = E.copy_query_codes(0, rev)
        assert np.all(body["score"] >= 3000)
        for h in body[:: max(1, body.size // 80)]:
            r = notrans["rcodes"][int(h["ref_start"]): int(h["ref_start"]) + int(h["len"]) + 1]
            q = qcodes[int(h["query_start"]): int(h["query_start"]) + int(h["len"]) + 1]
            raw = int(M[r, q].sum())
            assert (raw == int(h["score"])) if raw > 9000 else (int(h["score"]) <= raw)
