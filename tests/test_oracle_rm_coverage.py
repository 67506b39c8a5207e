"""CPU: the oracle's restatement of the repeat masker's post-processing (repeat_masker_src/seeder.cpp:153-188) against
an independent numpy model, and the interval plan (repeat_masker_src/main.cpp:316-436) against its invariants."""
import numpy as np
import pytest


def numpy_model(hsps, block_len, M):
    """difference array + cumsum, uint8 wrap, runs closed only by a following uncovered position."""
    d = np.zeros(block_len + 1, dtype=np.int64)
    for h in hsps:
        if h["len"] > 0:
            d[h["query_start"]] += 1
            d[h["query_start"] + h["len"]] -= 1
    depth = (np.cumsum(d)[:block_len] & 0xFF).astype(np.int64)
    cov = depth >= M
    out = []
    i = 0
    while i < block_len:
        if cov[i]:
            j = i
            while j < block_len and cov[j]:
                j += 1
            if j < block_len:  # no flush after the loop (:168-186)
                out.append((i, j - i))
            i = j
        else:
            i += 1
    return out


def random_hsps(O, rng, n, block_len, max_len):
    h = np.zeros(n, dtype=O.SEG_DTYPE)
    h["len"] = rng.integers(0, max_len, n)
    h["query_start"] = rng.integers(0, block_len - max_len, n)
    return h


@pytest.mark.parametrize("M", [0, 1, 2, 3, 7, 255, 256])
def test_coverage_runs_match_numpy_model(oracle, M):
    rng = np.random.default_rng(100 + M)
    for n in (0, 1, 5, 200):
        h = random_hsps(oracle, rng, n, 5000, 120)
        got = oracle.rm_coverage_intervals(h, 5000, M)
        assert [tuple(int(v) for v in r) for r in got] == numpy_model(h, 5000, M), (n, M)


def test_uint8_counters_wrap(oracle):
    """300 HSPs over the same stretch: depth 300 is stored as 44; with 256 copies the stretch reads as uncovered."""
    O = oracle
    h = np.zeros(256, dtype=O.SEG_DTYPE)
    h["query_start"] = 100
    h["len"] = 50
    assert O.rm_coverage_intervals(h, 1000, 1).size == 0
    h2 = np.concatenate([h, h[:44]])
    got = O.rm_coverage_intervals(h2, 1000, 44)
    assert [tuple(int(v) for v in r) for r in got] == [(100, 50)]
    assert O.rm_coverage_intervals(h2, 1000, 45).size == 0
    assert [tuple(int(v) for v in r) for r in got] == numpy_model(h2, 1000, 44)


def test_last_base_of_an_hsp_is_not_counted_and_open_runs_are_dropped(oracle):
    O = oracle
    h = np.zeros(2, dtype=O.SEG_DTYPE)
    h["query_start"] = [10, 90]
    h["len"] = [5, 9]  # bases 10..15 and 90..99 -> counted 10..14 and 90..98
    got = O.rm_coverage_intervals(h, 100, 1)
    assert [tuple(int(v) for v in r) for r in got] == [(10, 5), (90, 9)]
    h["len"] = [5, 10]  # a (malformed) HSP counted up to the last position of the block: the run stays open
    got = O.rm_coverage_intervals(h, 100, 1)
    assert [tuple(int(v) for v in r) for r in got] == [(10, 5)]


@pytest.mark.parametrize("seq_len,block,interval,prop", [
    (35_000_000, 1_000_000_000, 10_000_000, 0.2),
    (35_000_000, 1_000_000_000, 2_000_000, 0.3),
    (9_123_457, 4_000_000, 1_000_000, 0.5),
    (1_000, 1_000_000_000, 10_000_000, 0.2),
    (20_000_019, 1_000_000_000, 10_000_000, 1.0),
])
def test_plan_invariants(oracle, seq_len, block, interval, prop):
    tasks = oracle.rm_plan(seq_len, block, interval, prop, 19)
    assert tasks.size > 0
    covered = np.zeros(seq_len, dtype=bool)
    for t in tasks:
        bs, bl = int(t["block_start"]), int(t["block_len"])
        assert bs + bl <= seq_len
        assert t["start"] < t["end"] <= bl - 19
        assert t["end"] - t["start"] <= interval
        assert 0 <= t["ref_start"] <= t["start"] and t["end"] <= t["ref_end"] <= bl
        covered[bs + int(t["start"]):bs + int(t["end"])] = True
    # every position that can start a seed window inside its block is seeded exactly by some task
    per_block_end = {}
    for t in tasks:
        per_block_end[int(t["block_index"])] = max(per_block_end.get(int(t["block_index"]), 0), int(t["block_start"]) + int(t["end"]))
    assert covered[:max(per_block_end.values())].sum() >= covered.sum()
    assert covered.sum() >= seq_len - 19 * len(per_block_end) - 1
