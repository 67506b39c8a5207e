"""GPU: the engine driven with the reference's own call protocol -- every g_SendRefWriteRequest / GenerateSeedPosTable / g_ClearRef /
g_SendQueryWriteRequest / g_ClearQuery the source node of src/main.cpp makes (tests/golden/reader_golden.json: main.cpp:575-598, :603-735
verbatim, driven serially through src/seeder.cpp as it lies), replayed in its order: target blocks switched through g_ClearRef, both query buffers
cleared and refilled, a third query block taking over the buffer of a finished one.  At every payload the chunks of both strands go through
the drop-in and the device-seeded entry with the payload's buffer and must equal the oracle on that block pair."""
import numpy as np
import pytest

from segalign_amd import shard
from test_reader_golden import CASES, arenas

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_engine_under_the_reference_readers_protocol(oracle, engine, idx):
    E, O, c = engine, oracle, CASES[idx]
    R, Q = arenas(c)
    span, chunk, tr = len(c["shape"]), c["chunk"], bool(c["transition"])
    r_arr, q_arr = np.frombuffer(bytes(R.buf), dtype=np.uint8), np.frombuffer(bytes(Q.buf), dtype=np.uint8)
    sub_mat = O.build_sub_mat(910)
    hsps = checked = 0
    try:
        E.reset_option(None)
        E.InitializeInterface(1)
        k = E.GenerateShapePos(c["shape"])
        assert O.generate_shape_pos(c["shape"]) == k
        E.InitializeProcessor(tr, chunk, span, sub_mat, 910, 3000, False)
        keep, ref_codes, table, held = None, None, None, {}
        for e in c["events"]:
            if e[0] == "SendRef":
                keep = E.SendRefWriteRequest(r_arr, e[1], e[2])
                ref_codes = O.encode(bytes(R.buf[e[1]:e[1] + e[2]]))
            elif e[0] == "Table":
                E.GenerateSeedPosTable(keep, e[1], e[2], e[3], e[4], e[5])
                table = O.generate_seed_pos_table(bytes(R.buf), e[1], e[2], e[3], e[4], e[5])
            elif e[0] == "ClearRef":
                E.ClearRef()
            elif e[0] == "SendQuery":
                E.SendQueryWriteRequest(q_arr, e[1], e[2], e[3])
                blk = bytes(Q.buf[e[1]:e[1] + e[2]])
                held[e[3]] = (blk, bytes(Q.rc[e[1]:e[1] + e[2]]), O.encode_rev_comp(blk))
            elif e[0] == "ClearQuery":
                E.ClearQuery(e[1])
                del held[e[1]]
            elif e[0] == "Payload":
                q_len, s, t, buf = e[6], e[7], e[8], e[11]
                blk, rc_blk, (fw_codes, rc_codes) = held[buf]
                for rev in (False, True):
                    for (a, b) in shard.chunks_of((s, t), chunk, q_len, rev):
                        seeds = O.make_seeds(rc_blk if rev else blk, 0, a, b, span, k, tr)
                        if seeds.size == 0:
                            assert E.SeedAndFilterRange(a, b, rev, buf).size == 0
                            continue
                        want, _ = O.seed_and_filter(ref_codes, rc_codes if rev else fw_codes, table[0], table[1], seeds, sub_mat, span, 910, 3000, False)
                        for got in (E.SeedAndFilter(seeds, rev, buf), E.SeedAndFilterRange(a, b, rev, buf)):
                            assert np.array_equal(got, want), (idx, e, rev, a, b)
                        hsps += want.size - 1
                        checked += 1
        assert checked == sum(x[3] for x in c["events"] if x[0] == "SeedAndFilter")   # as many calls as the reference's seeder made
        assert hsps > 0
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)
