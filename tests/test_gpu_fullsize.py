"""GPU, BASELINE.json full size (configs[1] stand-in: 100 Mbp target, 250 kbp chunks): the oracle cannot run the
whole workload in seconds, so parity is checked through size-independent properties on a few calls, plus bit-exact
oracle agreement on full chunk calls, forty-chunk calls and repeat-masker intervals.  The oracle borrows NOTHING from the
device here: it builds its own seed position table from the ASCII target (orc_generate_seed_pos_table,
common/seed_pos_table.cu:49-109) and encodes both sequences itself (common/seed_filter_interface.cu:18-47,
src/seed_filter.cu:110-155); the device's tables and codes are held against those first."""
import numpy as np
import pytest

from helpers import check_seed_table_properties, pos_tables_equal_by_bucket
from segalign_amd import shard, synth

pytestmark = pytest.mark.gpu

SHAPE = "TTT0T00TT00T0T0TTTT"


@pytest.fixture(scope="module")
def full(oracle, engine, standin_100mbp):
    E, O = engine, oracle
    target, query = standin_100mbp
    sub_mat = O.build_sub_mat(910)
    E.InitializeInterface(1)
    k = E.GenerateShapePos(SHAPE)
    O.generate_shape_pos(SHAPE)
    E.InitializeProcessor(True, 250000, 19, sub_mat, 910, 3000, False)
    keep = E.SendRefWriteRequest(target, 0, target.size)
    E.GenerateSeedPosTable(keep, 0, target.size, 1, 19, k)
    E.SendQueryWriteRequest(query, 0, query.size, 0)
    # the oracle's OWN table and codes, from the ASCII (serial counting sort: ~20 s for 100 Mbp)
    o_index, o_pos = O.generate_seed_pos_table(target.tobytes(), 0, target.size, 1, 19, k)
    o_rcodes = O.encode(target.tobytes())
    o_q, o_qrc = O.encode_rev_comp(query.tobytes())
    yield dict(E=E, O=O, target=target, query=query, sub_mat=sub_mat, k=k, o_index=o_index, o_pos=o_pos, o_rcodes=o_rcodes, o_q=o_q, o_qrc=o_qrc)
    E.ShutdownProcessor()


def test_device_table_and_codes_equal_the_oracles_own_at_full_size(full):
    """a-5 / a-3 at workload size without properties standing in for equality: bucket ends word for word, positions bucket by bucket
    (canonical order inside a bucket, hazard H7), target codes, query codes of both strands."""
    E = full["E"]
    index, pos = E.copy_index_table(), E.copy_pos_table()
    assert index.size == full["o_index"].size == 1 << 24 and np.array_equal(index, full["o_index"])
    assert pos.size == full["o_pos"].size > 80_000_000
    assert pos_tables_equal_by_bucket(index, pos, full["o_pos"])   # (the oracle's buckets ascend; the device sorts every bucket its LDS holds)
    assert np.array_equal(E.copy_ref_codes(), full["o_rcodes"])
    assert np.array_equal(E.copy_query_codes(0, False), full["o_q"]) and np.array_equal(E.copy_query_codes(0, True), full["o_qrc"])


def test_table_is_a_permutation_of_valid_positions(full):
    check_seed_table_properties(full["E"], full["target"].size, 19)


def test_calls_are_deterministic_and_entry_points_agree(full):
    E, O, query = full["E"], full["O"], full["query"]
    qlen = query.size - 19
    rc_ascii = np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
    for rev, buf in ((False, query), (True, rc_ascii)):
        iv = (30_000_000, 40_000_000)
        (a, b) = shard.chunks_of(iv, 250000, qlen, rev)[3]
        out1 = E.SeedAndFilterRange(a, b, rev, 0)
        out2 = E.SeedAndFilterRange(a, b, rev, 0)
        seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, full["k"], True)
        out3 = E.SeedAndFilter(seeds, rev, 0)
        assert out1.size > 1 and np.array_equal(out1, out2) and np.array_equal(out1, out3)


def test_hsp_invariants_at_full_size(full):
    """Every returned HSP re-scored on the host from the encoded sequences gives its score (entropy factor 1 above
    3*hspthresh), passes the threshold, lies inside both sequences; header counts match; per-iteration order holds."""
    E, O = full["E"], full["O"]
    M = full["sub_mat"].reshape(8, 8)
    rcodes = E.copy_ref_codes()
    qlen = full["query"].size - 19
    checked = 0
    for rev in (False, True):
        qcodes = E.copy_query_codes(0, rev)
        for (a, b) in shard.chunks_of((50_000_000, 60_000_000), 250000, qlen, rev)[:3]:
            out = E.SeedAndFilterRange(a, b, rev, 0)
            st = E.last_call_stats()
            assert out[0]["len"] == out.size - 1 and out[0]["score"] == st["num_hits"] & 0x7FFFFFFF
            body = out[1:]
            assert np.all(body["score"] >= 3000)
            assert np.all(body["ref_start"].astype(np.int64) + body["len"] < rcodes.size)
            assert np.all(body["query_start"].astype(np.int64) + body["len"] < qcodes.size)
            # sorted by (query_start, ref_start, len, -score) inside each of the (at most 2) iterations
            key = body["query_start"].astype(np.int64) * (1 << 32) + body["ref_start"]
            assert np.count_nonzero(np.diff(key) < 0) <= 1
            for h in body[:: max(1, body.size // 60)]:
                r = rcodes[int(h["ref_start"]): int(h["ref_start"]) + int(h["len"]) + 1]
                q = qcodes[int(h["query_start"]): int(h["query_start"]) + int(h["len"]) + 1]
                raw = int(M[r, q].sum())
                if raw > 9000:
                    assert raw == int(h["score"])
                else:
                    assert int(h["score"]) <= raw
                assert M[r[0], q[0]] > 0 and M[r[-1], q[-1]] > 0  # an HSP starts and ends on its extreme prefix maxima
                checked += 1
    assert checked > 50


@pytest.mark.parametrize("rev", [False, True])
def test_one_full_chunk_bit_exact_vs_oracle(full, rev):
    E, O, query = full["E"], full["O"], full["query"]
    index, pos = full["o_index"], full["o_pos"]
    rcodes, qcodes = full["o_rcodes"], full["o_qrc"] if rev else full["o_q"]
    buf = query if not rev else np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
    a, b = 70_000_000, 70_250_000
    seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, full["k"], True)
    want, st = O.seed_and_filter(rcodes, qcodes, index, pos, seeds, full["sub_mat"])
    got = E.SeedAndFilter(seeds, rev, 0)
    assert st["num_hits"] > 5_000_000
    assert got.shape == want.shape and np.all(got == want)


def test_multi_chunk_calls_bit_exact_vs_oracle_at_full_size(full):
    """Half an interval (20 chunks per strand = one 20-chunk call of ~260 M hits per strand) through sa_seed_interval, every chunk
    against the oracle: the multi-chunk machinery (40 reference iterations in one pass, relative chain keys, per-segment LDS
    chains, speculative output copy) at the workload's real hit density."""
    E, O, query = full["E"], full["O"], full["query"]
    index, pos = full["o_index"], full["o_pos"]
    rcodes = full["o_rcodes"]
    qlen = query.size - 19
    iv = (40_000_000, 45_000_000)
    fw, rc, st = E.SeedInterval(iv[0], iv[1], qlen, E.STRAND_BOTH, 0, 2)
    hits = 0
    for rev, got in ((False, fw), (True, rc)):
        qcodes = full["o_qrc"] if rev else full["o_q"]
        buf = query if not rev else np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
        want = []
        for (a, b) in shard.chunks_of(iv, 250000, qlen, rev):
            seeds = O.make_seeds(buf.tobytes(), 0, a, b, 19, full["k"], True)
            w, ost = O.seed_and_filter(rcodes, qcodes, index, pos, seeds, full["sub_mat"])
            want.append(w[1:])
            hits += ost["num_hits"]
        want = np.concatenate(want)
        assert got.shape == want.shape and np.all(got == want), rev
    assert st["num_hits"] == hits and hits > 400_000_000 and fw.size + rc.size > 500


def test_the_bench_default_calls_bit_exact_vs_oracle_and_its_checksum(full):
    """What bench.py times, at the size and grouping it times it: interval 0 of the pass, both strands, through sa_seed_calls at the
    DEFAULT grouping (sa_get_chunks_per_call: two forty-chunk calls of ~520 M hits = 80 reference iterations each), every one of the
    80 chunks against the oracle; then the same interval through `bench.py --one-interval` in a process of its own: same calls,
    same HSP count, and a checksum equal to shard.hsp_checksum over the ORACLE's records -- the number an N-GPU run is verified by,
    pinned to the CPU restatement at workload size."""
    import json
    import os
    import subprocess
    import sys
    E, O, query = full["E"], full["O"], full["query"]
    index, pos = full["o_index"], full["o_pos"]
    rcodes = full["o_rcodes"]
    qlen = query.size - 19
    cpc = E.lib().sa_get_chunks_per_call()
    assert cpc == 40
    iv0 = shard.plan_intervals(query.size, 19, 10_000_000)[0]
    jobs = shard.call_jobs([iv0], qlen, 250000, cpc)
    assert [(j["chunks"], j["rev"]) for j in jobs] == [(40, False), (40, True)]
    hits = []
    outs, st = E.SeedCalls([(j["a"], j["b"], j["rev"]) for j in jobs], 0, 2, hits_out=hits)
    chk_oracle = chk_engine = n_hsps = 0
    for j, got in zip(jobs, outs):
        rev = j["rev"]
        qcodes = full["o_qrc"] if rev else full["o_q"]
        buf = query if not rev else np.frombuffer(O.rev_comp_ascii(query.tobytes(), 0, query.size), dtype=np.uint8)
        want, o_hits = [], 0
        for a in range(j["a"], j["b"], 250000):
            seeds = O.make_seeds(buf.tobytes(), 0, a, min(a + 250000, j["b"]), 19, full["k"], True)
            w, ost = O.seed_and_filter(rcodes, qcodes, index, pos, seeds, full["sub_mat"])
            want.append(w[1:])
            o_hits += ost["num_hits"]
        want = np.concatenate(want)
        assert got.shape == want.shape and np.all(got == want), rev
        assert o_hits == hits[len(hits) - len(jobs) + jobs.index(j)] and o_hits > 400_000_000
        chk_oracle += shard.hsp_checksum(want, rev)
        chk_engine += shard.hsp_checksum(got, rev)
        n_hsps += int(want.size)
    assert n_hsps > 500 and chk_engine == chk_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--one-interval", "--no-cpu-baseline", "--no-dropin"], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["calls"] == [[j["a"], j["b"], bool(j["rev"]), j["chunks"]] for j in jobs]
    assert line["hsps"] == n_hsps and line["hsp_checksum"] == chk_oracle % (1 << 55)


# ---- BASELINE configs[3]: repeat-masker path on the same 100 Mbp target (self-alignment, neighbor_proportion 0.2, M 1) ----
@pytest.fixture(scope="module")
def full_rm(full):
    """run_segalign_repeat_masker's engine state: the query IS the target (repeat_masker_src/seed_filter.cu:951-961)."""
    E, O, target = full["E"], full["O"], full["target"]
    E.RmSendQueryWriteRequest()
    rcodes = full["o_rcodes"]   # (the oracle's own codes and table: see the module docstring)
    yield dict(full, rcodes=rcodes, rc_codes=O.rev_comp_codes(rcodes),
               rc_ascii=np.frombuffer(O.rev_comp_ascii(target.tobytes(), 0, target.size), dtype=np.uint8),
               index=full["o_index"], pos=full["o_pos"])
    E.RmClearQuery()


def test_configs3_repeat_masker_plan_intervals_properties(full_rm):
    """Whole intervals of the reference's plan (repeat_masker_src/main.cpp:316-436) through sa_rm_mask_interval at full
    size: the runs are sorted, disjoint, non-empty, inside the block; totals are consistent; the call is deterministic."""
    E, O, target = full_rm["E"], full_rm["O"], full_rm["target"]
    L = target.size
    tasks = O.rm_plan(L, 1000000000, 10000000, 0.2, 19)
    assert tasks.size == 10 and int(tasks["block_len"][0]) == L
    seen = 0
    for t in (tasks[0], tasks[5], tasks[-1]):
        a, b, ws, we = int(t["start"]), int(t["end"]), int(t["ref_start"]), int(t["ref_end"])
        iv, tot = E.RmMaskInterval(a, b, ws, we, E.STRAND_BOTH, 1)
        iv2, tot2 = E.RmMaskInterval(a, b, ws, we, E.STRAND_BOTH, 1)
        assert np.array_equal(iv, iv2) and tot == tot2
        assert tot["num_seeds"] > 0 and tot["num_hits"] > tot["num_seeds"] and tot["num_hsps"] > 0
        s = iv["query_start"].astype(np.int64)
        e = s + iv["len"].astype(np.int64)
        assert np.all(iv["len"] > 0) and np.all(s >= 0) and np.all(e <= L)
        assert np.all(s[1:] > e[:-1])  # maximal runs: strictly separated, ascending
        # the trivial self-alignment covers every upper-case stretch of the interval that a seed reaches: the masked runs
        # lie inside the interval's own coordinates (plus strand) or anywhere in the window (minus strand)
        assert s.size > 100
        seen += s.size
    assert seen > 1000


@pytest.mark.parametrize("strands", [1, 2])
def test_configs3_repeat_masker_chunks_bit_exact_vs_oracle(full_rm, strands):
    """Two 250 kbp chunks of one interval, one strand per case: sa_rm_seed_and_filter (windowed SeedAndFilter, rm
    :724-876) and sa_rm_mask_interval (the whole seeder body) against the oracle on its own table."""
    E, O, target = full_rm["E"], full_rm["O"], full_rm["target"]
    L, chunk = target.size, 250000
    rev = strands == 2
    start_pos, end_pos, ws, we = 42_000_000, 42_500_000, 40_000_000, 52_000_000
    end_pos_rc = L - 1 - start_pos
    hsps = []
    for i in range(start_pos, end_pos, chunk):
        s0, s1 = i, min(i + chunk, end_pos)
        if rev:  # repeat_masker_src/seeder.cpp:118-119
            s0 = L - 1 - s1
            s1 = min(s0 + chunk, end_pos_rc)
        buf = full_rm["rc_ascii"] if rev else target
        seeds = O.make_seeds(buf.tobytes(), 0, s0, s1, 19, full_rm["k"], True)
        want, st = O.seed_and_filter(full_rm["rcodes"], full_rm["rc_codes"] if rev else full_rm["rcodes"], full_rm["index"],
                                     full_rm["pos"], seeds, full_rm["sub_mat"], rm=(rev, ws, we))
        got = E.RmSeedAndFilter(seeds, rev, ws, we)
        assert st["num_hits"] > 1_000_000
        assert got.shape == want.shape and np.all(got == want)
        hsps.append(want[1:])
    allh = np.concatenate(hsps)
    assert rev or allh.size > 0  # (the stand-in has no inverted repeats: the minus strand finds hits but no HSP here)
    want_iv = O.rm_coverage_intervals(allh, L, 1)
    got_iv, tot = E.RmMaskInterval(start_pos, end_pos, ws, we, strands, 1)
    assert np.array_equal(got_iv, want_iv) and tot["num_hsps"] == allh.size


def test_configs3_repeat_masker_grouped_interval_vs_oracle(full_rm):
    """Half an interval of the plan, both strands, through sa_rm_mask_interval -- twenty-chunk table-direct passes, a short last
    plus chunk whose minus chunk overlaps its neighbour (seeder.cpp:118-119) -- against the oracle chunk by chunk."""
    E, O, target = full_rm["E"], full_rm["O"], full_rm["target"]
    L, chunk = target.size, 250000
    start_pos, end_pos, ws, we = 42_000_000, 47_100_000, 40_000_000, 52_000_000
    end_pos_rc = L - 1 - start_pos
    hsps, seeds_n, hits_n = [], 0, 0
    for i in range(start_pos, end_pos, chunk):
        for rev in (False, True):
            s0, s1 = i, min(i + chunk, end_pos)
            if rev:
                s0 = L - 1 - s1
                s1 = min(s0 + chunk, end_pos_rc)
            buf = full_rm["rc_ascii"] if rev else target
            seeds = O.make_seeds(buf.tobytes(), 0, s0, s1, 19, full_rm["k"], True)
            if seeds.size == 0:
                continue
            want, st = O.seed_and_filter(full_rm["rcodes"], full_rm["rc_codes"] if rev else full_rm["rcodes"], full_rm["index"],
                                         full_rm["pos"], seeds, full_rm["sub_mat"], rm=(rev, ws, we))
            seeds_n += int(seeds.size)
            hits_n += int(st["num_hits"])
            hsps.append(want[1:])
    allh = np.concatenate(hsps)
    want_iv = O.rm_coverage_intervals(allh, L, 1)
    got_iv, tot = E.RmMaskInterval(start_pos, end_pos, ws, we, E.STRAND_BOTH, 1)
    assert np.array_equal(got_iv, want_iv)
    assert tot == dict(num_seeds=seeds_n, num_hits=hits_n, num_hsps=int(allh.size))
    assert hits_n > 400_000_000 and got_iv.size > 100
