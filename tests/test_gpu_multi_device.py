"""GPU, multi-device (the reference's own model: ONE process, all visible devices, calls handed to whichever device has a
free token -- common/seed_filter_interface.cu:49-80, src/seed_filter.cu:699-706,798-803; tables replicated on every device,
common/seed_pos_table.cu:33-47).

On a node with several GPUs these tests use them.  On a one-GPU box (the driver's test box) they still run: sa_select_devices
accepts an ordinal more than once, and every entry becomes an engine device of its own -- context, admin stream, target, seed
tables, table arena, four token-pool slots -- on that GPU.  Everything the multi-device code does differently from the
single-device code is exercised that way: N DevCtx, concurrent table builds (one host thread per engine device), the slot-major
token order, calls of one list landing on different engine devices, g_ClearRef + a block switch across contexts."""
import threading

import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import shard, synth

pytestmark = pytest.mark.gpu


def engine_devices(E, want=2):
    """Select `want` engine devices: distinct GPUs when the process sees that many, otherwise twins on ordinal 0."""
    import torch
    n = torch.cuda.device_count()
    ids = list(range(want)) if n >= want else [0] * want
    E.select_devices(ids)
    return ids


@pytest.fixture
def two_devices(engine):
    ids = engine_devices(engine, 2)
    yield ids
    engine.select_devices([])
    # the arenas are process-wide caches, one set per engine device: give the twins' memory back to the tests that follow
    engine.ShutdownProcessor()
    engine.ReleaseArena()


def test_all_devices_token_pool_matches_oracle(oracle, engine, two_devices):
    E = engine
    t, q = synth.make_pair(300000, 13, 14, sub_rate=0.1, mask_frac=0.1, records=2, indel_every=600)
    c = Case(t, q, chunk=25000).oracle_setup(oracle)
    c.engine_setup(E, num_gpu=-1)  # every device gets target, tables and both query strands (seed_pos_table.cu:33-47)
    try:
        jobs = [(rev, s, e) for rev in (False, True) for (s, e) in c.chunks()]
        want = {j: c.oracle_saf(c.host_seeds(j[1], j[2], j[0]), j[0])[0] for j in jobs}
        got, got_dev, devices = {}, {}, set()
        lock = threading.Lock()

        def work(my):
            for j in my:
                a = E.SeedAndFilter(c.host_seeds(j[1], j[2], j[0]), j[0], 0)
                d1 = E.last_call_stats()["device"]
                b = E.SeedAndFilterRange(j[1], j[2], j[0], 0)
                d2 = E.last_call_stats()["device"]
                with lock:
                    got[j], got_dev[j] = a, b
                    devices.update((d1, d2))

        threads = [threading.Thread(target=work, args=(jobs[i::8],)) for i in range(8)]
        [th.start() for th in threads]
        [th.join() for th in threads]
        assert all(seg_equal(got[j], want[j]) and seg_equal(got_dev[j], want[j]) for j in jobs)
        assert len(devices) >= 2, devices  # the pool really spread the calls
        for d in range(len(two_devices)):  # replicated state is identical on every device
            assert np.array_equal(E.copy_index_table(d), c.o_index) and np.array_equal(E.copy_pos_table(d), c.o_pos)
    finally:
        E.ShutdownProcessor()


def _pass(E, jobs, threads):
    devs = []
    outs, st = E.SeedCalls([(j["a"], j["b"], j["rev"]) for j in jobs], 0, threads, devices_out=devs)
    return outs, st, devs


def test_call_list_over_two_devices_equals_one_device_and_survives_a_block_switch(oracle, engine, two_devices):
    """20 Mbp block pair, multi-chunk calls through the engine's worker pool (what bench.py and segalign_host issue): the HSPs of
    every call from two engine devices equal those from one, both devices take calls, and after g_ClearRef + a new target block
    (tables rebuilt concurrently on both) the same holds again."""
    E = engine
    sub_mat = oracle.build_sub_mat(910)
    blocks = [synth.make_pair(20_000_000, 31, 32, sub_rate=0.08, mask_frac=0.2, records=3, invert_frac=0.3, invert_block=100_000),
              synth.make_pair(12_000_000, 33, 34, sub_rate=0.10, mask_frac=0.1, records=2, invert_frac=0.2, invert_block=50_000)]
    shape = "TTT0T00TT00T0T0TTTT"

    def run(ids):
        E.select_devices(ids)
        E.InitializeInterface(len(ids))
        k = E.GenerateShapePos(shape)
        E.InitializeProcessor(True, 250_000, 19, sub_mat, 910, 3000, False)
        res = []
        for (t, q) in blocks:
            keep = E.SendRefWriteRequest(t, 0, t.size)
            E.GenerateSeedPosTable(keep, 0, t.size, 1, 19, k)
            E.SendQueryWriteRequest(q, 0, q.size, 0)
            ivs = shard.plan_intervals(q.size, 19, 10_000_000)
            jobs = shard.call_jobs(ivs, q.size - 19, 250_000, 10)
            outs, st, devs = _pass(E, jobs, 6)
            tables = [(E.copy_index_table(d), E.copy_pos_table(d)) for d in range(len(ids))]
            res.append((outs, st, devs, tables))
            E.ClearQuery(0)
            E.ClearRef()
        E.ShutdownProcessor()
        return res

    try:
        one = run(two_devices[:1])
        two = run(two_devices)
    finally:
        E.select_devices([])
    for b in range(len(blocks)):
        o1, s1, d1, t1 = one[b]
        o2, s2, d2, t2 = two[b]
        assert len(o1) == len(o2) and sum(o.size for o in o1) > 1000
        assert all(seg_equal(a, c) for a, c in zip(o1, o2))
        assert s1["num_hits"] == s2["num_hits"] and s1["num_anchors"] == s2["num_anchors"]
        assert set(d1) == {0} and set(d2) == {0, 1}, (set(d1), set(d2))
        # both devices carry the same tables as the single-device run
        for (ix, ps) in t2:
            assert np.array_equal(ix, t1[0][0]) and np.array_equal(ps, t1[0][1])


def test_hit_counts_by_lookup_equal_the_calls_own_counts(oracle, engine, two_devices):
    """sa_count_call_hits (the cheap weighting pass of a multi-GPU host) reports exactly the hits the calls themselves see."""
    E = engine
    sub_mat = oracle.build_sub_mat(910)
    t, q = synth.make_pair(6_000_000, 41, 42, sub_rate=0.08, mask_frac=0.2, records=2, invert_frac=0.3, invert_block=100_000)
    try:
        E.InitializeInterface(len(two_devices))
        k = E.GenerateShapePos("TTT0T00TT00T0T0TTTT")
        E.InitializeProcessor(True, 250_000, 19, sub_mat, 910, 3000, False)
        keep = E.SendRefWriteRequest(t, 0, t.size)
        E.GenerateSeedPosTable(keep, 0, t.size, 1, 19, k)
        E.SendQueryWriteRequest(q, 0, q.size, 0)
        jobs = shard.call_jobs(shard.plan_intervals(q.size, 19, 10_000_000), q.size - 19, 250_000, 8)
        calls = [(j["a"], j["b"], j["rev"]) for j in jobs]
        counted = E.CountCallHits(calls, 0, 4)
        seen = []
        E.SeedCalls(calls, 0, 4, hits_out=seen)
        assert counted == seen and sum(counted) > 0
        # ... and per 250 kbp chunk: the same lookups report every chunk's hits, which add up to the calls'
        per_chunk = E.CountCallHits(calls, 0, 4, per_chunk=True)
        assert len(per_chunk) == sum(j["chunks"] for j in jobs) and sum(per_chunk) == sum(counted)
        singles = [(a, min(a + 250_000, j["b"]), j["rev"]) for j in jobs for a in range(j["a"], j["b"], 250_000)]
        assert per_chunk == E.CountCallHits(singles, 0, 4)
    finally:
        E.ShutdownProcessor()


def test_eight_engine_devices_times_six_slots_on_one_node(oracle, engine):
    """What the driver's 8-GPU run instantiates in ONE process under the reference's own model (all visible devices behind one token pool,
    common/seed_filter_interface.cu:49-80): 8 engine devices x 6 slots = 48 tokens, 48 streams + 8 upload streams, 48 engine worker
    threads, 8 table arenas and 8 work arenas, 8 concurrent table builds.  On a one-GPU box the eight are twins on ordinal 0 (arenas
    kept small: 8 x the default 40 + 18 GiB would not fit one GPU); on an 8-GPU node they are the eight GPUs.  Checked: every device
    holds the oracle's table, every device takes calls, the call list's HSPs equal the one-device run's and the oracle's."""
    E = engine
    ids = engine_devices(E, 8)
    E.set_option("slots", 6)
    E.set_option("arena_gb", 2)
    E.set_option("work_gb", 1)
    t, q = synth.make_pair(4_000_000, 51, 52, sub_rate=0.09, mask_frac=0.15, records=3, invert_frac=0.3, invert_block=50_000, indel_every=700)
    c = Case(t, q, chunk=50_000).oracle_setup(oracle)
    try:
        c.engine_setup(E, num_gpu=-1)
        for d in range(8):
            assert np.array_equal(E.copy_index_table(d), c.o_index) and np.array_equal(E.copy_pos_table(d), c.o_pos)
        ivs = shard.plan_intervals(q.size, 19, 1_000_000)
        jobs = shard.call_jobs(ivs, q.size - 19, 50_000, 2)            # 2-chunk calls: ~80 calls for 48 slots
        devs, hits = [], []
        outs, st = E.SeedCalls([(j["a"], j["b"], j["rev"]) for j in jobs], 0, 48, hits_out=hits, devices_out=devs)
        assert len(jobs) >= 64 and set(devs) == set(range(8)), sorted(set(devs))
        # the oracle, chunk by chunk
        n = 0
        for j, got in zip(jobs, outs):
            want = [c.oracle_saf(c.host_seeds(a, min(a + 50_000, j["b"]), j["rev"]), j["rev"])[0][1:] for a in range(j["a"], j["b"], 50_000)]
            want = np.concatenate(want)
            assert seg_equal(got, want), (j["a"], j["b"], j["rev"])
            n += want.size
        assert n > 200
        # 48 HOST threads through the drop-in and the device-seeded entry at once (the reference's TBB seeder bodies, src/seeder.cpp:78)
        chunks = [(rev, s, e) for rev in (False, True) for (s, e) in c.chunks()]
        want1 = {j: c.oracle_saf(c.host_seeds(j[1], j[2], j[0]), j[0])[0] for j in chunks[:96]}
        bad, seen = [], set()
        lock = threading.Lock()

        def work(my):
            for j in my:
                a = E.SeedAndFilter(c.host_seeds(j[1], j[2], j[0]), j[0], 0)
                d = E.last_call_stats()["device"]
                b = E.SeedAndFilterRange(j[1], j[2], j[0], 0)
                with lock:
                    seen.add(d)
                    if not (seg_equal(a, want1[j]) and seg_equal(b, want1[j])):
                        bad.append(j)

        ths = [threading.Thread(target=work, args=(list(want1)[i::48],)) for i in range(48)]
        [th.start() for th in ths]
        [th.join() for th in ths]
        assert not bad and len(seen) >= 4, (bad[:3], seen)
    finally:
        E.ShutdownProcessor()
        E.select_devices([])
        E.reset_option(None)
        E.ReleaseArena()
