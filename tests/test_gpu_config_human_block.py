"""GPU, BASELINE.json configs[2] at the size that matters to the engine: ONE reference-sized target block.  The reference
closes a block once it exceeds 500 Mbp (src/graph.h:10, src/main.cpp:359,515), so a human-scale run is 6 such blocks;
everything per block is what one GPU holds at a time (pos_table ~2 GB, ~30 hits per seed word, >60 M hits per 250 kbp
call).  The oracle cannot build a 500 Mbp table or run whole intervals in seconds, so: table properties at full size
(permutation of the valid positions, ascending buckets, keys consistent on samples) and bit-exact chunk calls on both
strands with the table copied from the device."""
import numpy as np
import pytest

from segalign_amd import synth

pytestmark = pytest.mark.gpu

SHAPE = "TTT0T00TT00T0T0TTTT"
TLEN = 500_000_000


@pytest.fixture(scope="module")
def human_block(oracle, engine):
    E, O = engine, oracle
    t = synth.random_dna(TLEN, 5)
    t = synth.soft_mask(t, 6, 0.3, 200, 2000)
    per = TLEN // 4
    target = synth.join_records([t[i * per:(i + 1) * per] for i in range(4)])
    del t
    # query block: 4 Mbp of 1.2 %-diverged pieces of distant target regions, every third one inverted (block shuffles)
    rng = np.random.default_rng(7)
    pieces = []
    for i in range(16):
        p = int(rng.integers(0, target.size - 300000))
        seg = synth.mutate(target[p:p + 250000].copy(), 100 + i, 0.012, indel_every=900)
        pieces.append(synth.reverse_complement(seg) if i % 3 == 0 else seg)
    query = np.concatenate(pieces)
    sub_mat = O.build_sub_mat(910)
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    O.generate_shape_pos(SHAPE)
    E.InitializeProcessor(True, 250000, 19, sub_mat, 910, 3000, False)
    keep = E.SendRefWriteRequest(target, 0, target.size)
    E.GenerateSeedPosTable(keep, 0, target.size, 1, 19, k)
    E.SendQueryWriteRequest(query, 0, query.size, 0)
    yield dict(E=E, O=O, target=target, query=query, sub_mat=sub_mat, k=k, index=E.copy_index_table(), pos=E.copy_pos_table(),
               rcodes=E.copy_ref_codes())
    E.ShutdownProcessor()


def test_configs2_block_table_is_a_permutation_of_valid_positions(human_block):
    O, target = human_block["O"], human_block["target"]
    index, pos, codes = human_block["index"], human_block["pos"], human_block["rcodes"]
    n = pos.size
    assert int(index[-1]) == n and n > 100_000_000
    assert np.all(index[1:] >= index[:-1])
    assert int(pos.min()) >= 1 and int(pos.max()) <= target.size - 19       # position 0 is never indexed (H6)
    seen = np.zeros(target.size, dtype=bool)
    seen[pos] = True
    assert int(np.count_nonzero(seen)) == n                                  # all distinct
    bad = np.concatenate([[0], np.cumsum(codes >= 4, dtype=np.int32)])
    valid = (bad[19:] - bad[:-19]) == 0                                      # windows of upper-case ACGT only
    valid[0] = False
    assert np.array_equal(valid, seen[:valid.size])                          # exactly the valid windows are indexed
    del seen, valid, bad
    # ascending inside every bucket: descents only at bucket starts
    desc = np.flatnonzero(pos[1:] < pos[:-1]) + 1
    is_start = np.zeros(n + 1, dtype=bool)
    is_start[index[:-1]] = True
    is_start[0] = True
    assert np.all(is_start[desc])
    # keys: sampled buckets hold positions whose spaced k-mer IS the bucket's key (ntcoding.cpp:43-61)
    rng = np.random.default_rng(1)
    tb = target.tobytes()
    for key in rng.integers(0, index.size, 300):
        b = int(index[key - 1]) if key else 0
        for p in pos[b:int(index[key])][:3]:
            assert O.kmer_index_at_pos(tb[int(p):int(p) + 19], 0, 19) == int(key)


@pytest.mark.parametrize("rev", [False, True])
def test_configs2_block_chunks_bit_exact_vs_oracle(human_block, rev):
    E, O, query = human_block["E"], human_block["O"], human_block["query"]
    qcodes = E.copy_query_codes(0, rev)
    buf = query if not rev else np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
    end_pos = query.size - 19
    total_hits, n_hsps = 0, 0
    for a in (0, 250000):
        b = min(a + 250000, end_pos)
        seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, human_block["k"], True)
        want, st = O.seed_and_filter(human_block["rcodes"], qcodes, human_block["index"], human_block["pos"], seeds,
                                     human_block["sub_mat"])
        got = E.SeedAndFilterRange(a, b, rev, 0)
        assert got.shape == want.shape and np.all(got == want), (rev, a, b, got[:3], want[:3])
        total_hits += st["num_hits"]
        n_hsps += want.size - 1
    assert total_hits > 60_000_000 and n_hsps > 0  # (one of the two pieces is an inversion: its HSPs are on the other strand)
    # the multi-chunk entry over the same stretch (two chunks in one pass: > 100 M hits in one launch)
    outs = E.SeedAndFilterChunks(0, 500000, rev, 0)
    assert np.array_equal(outs[0], E.SeedAndFilterRange(0, 250000, rev, 0))
    assert np.array_equal(outs[1], E.SeedAndFilterRange(250000, 500000, rev, 0))


# MAX_HITS of the GPUs the reference was run on (README.md:27; src/seed_filter.cu:832-841): 33.5 M on an 8 GiB M60, 66.2 M on a 16 GB V100,
# against > 60 M hits per 250 kbp chunk of a 500 Mbp block -- see tests/test_gpu_config_lumpy.py.
@pytest.mark.parametrize("name,mem", [("M60_8GiB", 8 << 30), ("V100_16GB", 16945512448)], ids=["M60_8GiB", "V100_16GB"])
def test_configs2_heaviest_chunks_under_the_reference_gpus_max_hits(human_block, name, mem):
    """The heaviest chunk of each strand of the query block and its two neighbours under the reference GPUs' MAX_HITS: single-chunk
    device-seeded calls and the grouped entry against the oracle with the same MAX_HITS (iteration plans of :718-745 at human density)."""
    E, O, query = human_block["E"], human_block["O"], human_block["query"]
    mh = E.max_hits_for_mem(mem)
    assert mh == O.max_hits_for_mem(mem)
    end_pos = query.size - 19
    E.set_max_hits(mh)
    try:
        split = 0
        for rev in (False, True):
            ch = [(a, min(a + 250000, end_pos), rev) for a in range(0, end_pos, 250000)]
            hits = np.array(E.CountCallHits(ch, 0, 4), dtype=np.int64)
            i = int(np.argmax(hits))
            lo = max(0, min(i - 1, len(ch) - 3))
            group = ch[lo:lo + 3]
            qcodes = E.copy_query_codes(0, rev)
            buf = query if not rev else np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
            want = []
            for j, (a, b, _) in enumerate(group):
                seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, human_block["k"], True)
                w, ost = O.seed_and_filter(human_block["rcodes"], qcodes, human_block["index"], human_block["pos"], seeds, human_block["sub_mat"],
                                           max_hits=mh)
                assert ost["num_hits"] == hits[lo + j]
                got = E.SeedAndFilterRange(a, b, rev, 0)
                st = E.last_call_stats()
                assert got.shape == w.shape and np.all(got == w), (name, rev, a, b, got.size, w.size)
                assert st["num_iter"] == ost["num_iter"] and st["lookup_path"] == 2 and not (st["path_flags"] & E.PATH_GENERAL_FALLBACK)
                split += int(ost["num_hits"] >= mh)
                want.append(w[1:])
            outs, _ = E.SeedCalls([(group[0][0], group[-1][1], rev)], 0, 1)
            w_all = np.concatenate(want)
            assert outs[0].shape == w_all.shape and np.all(outs[0] == w_all), (name, "grouped", rev)
        assert (split == 6) if mem == 8 << 30 else (split < 6), (name, split)
    finally:
        E.set_max_hits(0)


def test_configs2_block_multi_chunk_call_bit_exact_vs_oracle(human_block):
    """ONE call over sixteen chunks of the plus strand at human-scale hit density (~0.5 G hits, > 1 M chain candidates, tens
    of thousands of survivors in 32 dedup segments): every chunk's vector against the oracle."""
    E, O, query = human_block["E"], human_block["O"], human_block["query"]
    qcodes = E.copy_query_codes(0, False)
    end_pos = query.size - 19
    assert end_pos > 15 * 250000
    outs = E.SeedAndFilterChunks(0, min(16 * 250000, end_pos), False, 0)
    st_call = E.last_call_stats()
    hits = 0
    for c in range(16):
        a, b = c * 250000, min((c + 1) * 250000, end_pos)
        seeds = O.make_seeds(query.tobytes(), 0, a, b, 19, human_block["k"], True)
        want, st = O.seed_and_filter(human_block["rcodes"], qcodes, human_block["index"], human_block["pos"], seeds,
                                     human_block["sub_mat"])
        assert outs[c].shape == want.shape and np.all(outs[c] == want), (c, outs[c][:3], want[:3])
        hits += st["num_hits"]
    assert st_call["num_hits"] == hits and hits > 400_000_000
