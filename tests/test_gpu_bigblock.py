"""GPU, scale: a 750 Mbp target block -- 1.5x the size at which the reference closes a block (src/main.cpp:320-549: 500 Mbp), 7.2 G
neighbourhood-table entries = 230 GB of context records, run offsets far beyond 2^32 -- against the oracle on whole 250 kbp chunks of
both strands.  Every chunk compared holds homologous pieces in BOTH orientations, so both strands return HSPs (a check on two chunks
with no anchors says nothing about the records).  Reference: src/seed_filter.cu:157-230 (lookup / hits), :232-652 (find_hsps),
:682-828 (SeedAndFilter)."""
import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def test_750_mbp_block_both_strands_against_the_oracle(oracle, engine):
    E = engine
    E.ShutdownProcessor()
    E.ReleaseArena()   # (whatever earlier modules left cached: this block needs 230 GB)
    tlen = 750_000_000
    target = synth.random_dna(tlen, 5)
    target = synth.soft_mask(target, 6, 0.3, 200, 2000)
    target = synth.join_records([target[i:i + tlen // 4] for i in range(0, tlen, tlen // 4)][:4])
    # query block: 1 Mbp of 125 kbp pieces of distant target regions (some from beyond 2^29 and 2^29.4 in the block), diverged 2-5.5 %,
    # sparse indels, alternately as they are and inverted: every 250 kbp chunk of either strand holds one piece of each orientation
    rng = np.random.default_rng(7)
    starts = [int(x) for x in rng.integers(0, target.size - 200_000, 8)]
    starts[1], starts[2] = 700_000_000, 560_000_123
    pieces = []
    for i, p in enumerate(starts):
        seg = synth.mutate(target[p:p + 125_000].copy(), 100 + i, 0.02 + 0.005 * i, indel_every=900)
        pieces.append(synth.reverse_complement(seg) if i % 2 else seg)
    query = np.concatenate(pieces)
    c = Case(target, query).oracle_setup(oracle)
    try:
        c.engine_setup(E)
        assert E.lookup_mode() == 2 and E.neighbourhood_entries() > (1 << 32)
        assert np.array_equal(E.copy_index_table(), c.o_index)
        wants = {}
        for rev in (False, True):
            for (s, e) in c.chunks()[:2]:
                got = E.SeedAndFilterRange(s, e, rev, 0)
                st = E.last_call_stats()
                want, _ = c.oracle_saf(c.host_seeds(s, e, rev), rev)
                wants[(rev, s)] = want
                assert want.size - 1 >= 20, (rev, s, want.size)   # the chunk really has HSPs on this strand
                assert st["num_hits"] > 50_000_000 and st["lookup_path"] == 2
                assert seg_equal(got, want), (rev, s, e, got.size, want.size)
        # one multi-chunk call over the whole strand = the four chunks one by one
        for rev in (False, True):
            ch = c.chunks()
            outs = E.SeedAndFilterChunks(ch[0][0], ch[-1][1], rev, 0)
            for j, (s, e) in enumerate(ch[:2]):
                assert seg_equal(outs[j], wants[(rev, s)])
    finally:
        E.ShutdownProcessor()
        E.ReleaseArena()   # 230 GB of table arena: give it back before the next module
