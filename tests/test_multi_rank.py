"""CPU-only: the N>1 path of bench.py (one process per GPU, interval shards, no data-path collective) rehearsed with
world_size 2 over gloo, launched exactly like the driver launches it (torch.distributed.run, 127.0.0.1)."""
import json
import os
import subprocess
import sys

from segalign_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interval_and_chunk_plan_matches_reference_rules():
    # src/main.cpp:383-393: intervals tile [0, len - seed_size) ; seeder.cpp:48-51: chunks tile the interval
    ivs = shard.plan_intervals(25_000_019, 19, 10_000_000)
    assert ivs == [(0, 10_000_000), (10_000_000, 20_000_000), (20_000_000, 25_000_000)]
    fw = shard.chunks_of(ivs[2], 250_000, 25_000_000, False)
    assert fw[0] == (20_000_000, 20_250_000) and fw[-1] == (24_750_000, 25_000_000) and len(fw) == 20
    # minus strand: same interval in rc coordinates (seeder.cpp:33-34)
    rc = shard.chunks_of(ivs[2], 250_000, 25_000_000, True)
    assert rc[0] == (0, 250_000) and rc[-1][1] == 5_000_000
    # shards partition the interval list
    parts = [shard.shard(ivs, r, 2) for r in range(2)]
    assert sorted(parts[0] + parts[1]) == ivs and not set(parts[0]) & set(parts[1])


def _run(nproc, steps, scaling="strong", port=29541, mbp=60, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", "0",
            "--dry-run", "--target-mbp", str(mbp), "--scaling", scaling] + list(extra)
    if nproc == 1:
        cmd = base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + base[1:]
    out = subprocess.check_output(cmd, env=env, stderr=subprocess.DEVNULL, timeout=300).decode().strip().split("\n")
    return json.loads(out[-1])


def _stub(j, chunk, w=1):
    return sum(w * ((a * 31 + min(a + chunk, j["b"]) * 17 + int(j["rev"])) % 1000003) for a in range(j["a"], j["b"], chunk))


def _model(world, steps, scaling, qlen=60_000_000, interval=10_000_000, chunk=250_000):
    """Independent statement of the bench's work map.  strong: the calls of a pass (per strand the 250 kbp chunks of
    src/seeder.cpp:48-51 / :33-34 in equal groups of <= 16 consecutive full-sized chunks, plus strand first) are dealt to the ranks,
    each exactly once per step; weak: in step k rank r walks the whole call list from call r + k on, wrapping."""
    ivs = shard.plan_intervals(qlen, 19, interval)
    jobs = []
    L = qlen - 19
    for rev in (False, True):
        # the strand's chunk starts in ascending order; a range can only continue while chunks touch and are full-sized
        spans = [(L - e, L - s) for (s, e) in reversed(ivs)] if rev else list(ivs)
        run = []          # (a, b) of consecutive chunks one call range can describe
        runs = []
        for (lo, hi) in spans:
            c = lo
            while c < hi:
                d = min(c + chunk, hi)
                if run and (run[-1][1] != c or run[-1][1] - run[-1][0] != chunk):
                    runs.append(run)
                    run = []
                run.append((c, d))
                c = d
        runs.append(run)
        for run in runs:
            n = len(run)
            calls = -(-n // 16)
            group = -(-n // calls)
            for c in range(0, n, group):
                jobs.append(dict(rev=rev, a=run[c][0], b=run[min(c + group, n) - 1][1]))
    bases = check = 0
    for r in range(world):
        for k in range(steps):
            if scaling == "strong":
                todo = [(1, j) for i, j in enumerate(jobs) if i % world == r]
            else:
                todo = [(i + 1, jobs[(r + k + i) % len(jobs)]) for i in range(len(jobs))]
            for w, j in todo:
                bases += 0 if j["rev"] else j["b"] - j["a"]
                check += _stub(j, chunk, w)
    return bases, check


def test_call_list_partition_covers_every_call_exactly_once():
    ivs = shard.plan_intervals(60_000_000, 19, 10_000_000)
    jobs = shard.call_jobs(ivs, 60_000_000 - 19, 250_000, 16)
    assert sum(j["chunks"] for j in jobs) == 2 * 240 and max(j["chunks"] for j in jobs) <= 16
    assert len(jobs) == 15 + 3 + 13   # plus: 240 chunks in 15 calls; minus: the short tail interval (40 chunks: 14 + 14 + 12), then 200 in 13
    # every strand is tiled by its calls without gap or overlap, and every call range cuts into full chunks except its last
    for rev in (False, True):
        mine = sorted((j["a"], j["b"]) for j in jobs if j["rev"] == rev)
        assert mine[0][0] == 0 and mine[-1][1] == 60_000_000 - 19 and all(mine[i][1] == mine[i + 1][0] for i in range(len(mine) - 1))
        want = [(a, b) for (_, a, b) in shard.strand_chunks(ivs, 60_000_000 - 19, 250_000, rev)]
        got = [(c, min(c + 250_000, b)) for (a, b) in mine for c in range(a, b, 250_000)]
        assert got == want   # the engine's cut of every call range reproduces the reference's chunk grid (seeder.cpp:48-51)
    for world in (1, 2, 3, 8):
        parts = [shard.partition(jobs, r, world) for r in range(world)]
        flat = [id(j) for p in parts for j in p]
        assert sorted(flat) == sorted(id(j) for j in jobs)  # exactly once
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        # weighted map (seed hits per call in the real bench): still every call exactly once, original order inside a rank,
        # and the heaviest rank carries at most one call's weight more than the lightest
        rng = __import__("random").Random(7 + world)
        w = [rng.randint(1, 1000) * (5 if k % 7 == 0 else 1) for k in range(len(jobs))]
        wparts = [shard.partition(jobs, r, world, w) for r in range(world)]
        assert sorted(id(j) for p in wparts for j in p) == sorted(id(j) for j in jobs)
        pos = {id(j): k for k, j in enumerate(jobs)}
        loads = []
        for p in wparts:
            ks = [pos[id(j)] for j in p]
            assert ks == sorted(ks)
            loads.append(sum(w[k] for k in ks))
        assert max(loads) - min(loads) <= max(w)


def test_the_deal_continues_from_pass_to_pass_and_evens_out():
    """20 calls on 8 ranks are 3/3/3/3/2/2/2/2 in one pass; with the deal continuing (offset = calls dealt so far) every pass still
    deals every call exactly once and the shares are even over two passes -- the default map of the timed region."""
    jobs = [dict(a=i, b=i + 1, rev=False, chunks=1, interval=0) for i in range(20)]
    total = [0] * 8
    for k in range(4):
        parts = [shard.partition(jobs, r, 8, None, offset=k * len(jobs)) for r in range(8)]
        assert sorted(j["a"] for p in parts for j in p) == list(range(20))      # every call exactly once per pass
        assert max(len(p) for p in parts) - min(len(p) for p in parts) == 1
        for r in range(8):
            total[r] += len(parts[r])
        if k % 2 == 1:
            assert max(total) == min(total)                                       # even after every second pass
    assert shard.partition(jobs, 3, 8, None) == shard.partition(jobs, 3, 8, None, offset=0)


def test_strong_scaling_partition_keeps_the_checksum_for_world_1_2_3():
    """The strong-scaling map: total work fixed, every call on exactly one rank, so bases and the stub checksum of a pass do not
    depend on the number of ranks -- and equal the independent model."""
    one = _run(1, 2)
    two = _run(2, 2, port=29541)
    three = _run(3, 2, port=29543)
    assert one["scaling"] == two["scaling"] == three["scaling"] == "strong"
    assert (one["bases"], one["checksum"]) == _model(1, 2, "strong")
    assert (two["bases"], two["checksum"]) == (one["bases"], one["checksum"])
    assert (three["bases"], three["checksum"]) == (one["bases"], one["checksum"])
    assert two["n_gpus"] == 2 and three["n_gpus"] == 3


def test_weak_scaling_rank_walks_and_reduction_match_the_model():
    # the stub checksum is weighted by the walk position, so a wrong rank walk, chunk bound or sum / max reduction changes the line
    one = _run(1, 2, "weak")
    two = _run(2, 2, "weak", port=29545)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert (one["bases"], one["checksum"]) == _model(1, 2, "weak")
    assert (two["bases"], two["checksum"]) == _model(2, 2, "weak")
    assert two["bases"] == 2 * one["bases"]  # weak scaling: per-GPU work is fixed


def test_world_8_on_the_human_deal_is_even_and_keeps_the_checksum():
    """The driver's N = 8 launch on configs[2]'s call list: one 500 Mbp query block in 15-chunk calls = 268 calls per pass (what
    sa_get_chunks_per_call() gives a 500 Mbp target block: tools/human_grid.py), dealt round-robin with the deal continuing from pass
    to pass.  Eight gloo ranks, launched as the driver launches them: every call exactly once per pass (checksum and bases equal the
    one-rank run's), and no rank carries more than 1.04 x the mean share."""
    extra = ["--chunks-per-call", "15"]
    one = _run(1, 2, mbp=500, extra=extra)
    eight = _run(8, 2, port=29549, mbp=500, extra=extra)
    assert one["config"]["calls_per_step"] == eight["config"]["calls_per_step"] == 268
    assert eight["n_gpus"] == 8 and (eight["bases"], eight["checksum"]) == (one["bases"], one["checksum"])
    assert one["bases"] == 2 * (500_000_000 - 19)
    assert 1.0 <= eight["config"]["partition_imbalance"] <= 1.04, eight["config"]["partition_imbalance"]
