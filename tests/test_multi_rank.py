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


def _run(nproc, steps):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", "0",
            "--dry-run", "--target-mbp", "60"]
    if nproc == 1:
        cmd = base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", "29541"] + base[1:]
    out = subprocess.check_output(cmd, env=env, stderr=subprocess.DEVNULL, timeout=300).decode().strip().split("\n")
    return json.loads(out[-1])


def _model(world, steps, qlen=60_000_000, interval=10_000_000, chunk=250_000):
    """Independent statement of the bench's work map: in step k rank r walks the interval list from interval r + k on,
    wrapping (weak scaling: every rank covers every interval once per step); chunks per src/seeder.cpp:48-51,:33-34."""
    ivs = shard.plan_intervals(qlen, 19, interval)
    bases = check = 0
    for r in range(world):
        for k in range(steps):
            for i in range(len(ivs)):
                iv = ivs[(r + k + i) % len(ivs)]
                bases += iv[1] - iv[0]
                for rev in (False, True):
                    s, e = (qlen - 19 - iv[1], qlen - 19 - iv[0]) if rev else iv
                    for a in range(s, e, chunk):
                        check += (i + 1) * ((a * 31 + min(a + chunk, e) * 17 + int(rev)) % 1000003)
    return bases, check


def test_rank_walks_and_reduction_match_the_model():
    # 60 Mbp -> 6 intervals; the stub engine's checksum depends on WHICH interval a rank visits WHEN, so a wrong
    # interval-to-rank map, a wrong chunk bound or a wrong sum/max reduction changes the line
    one = _run(1, 2)
    two = _run(2, 2)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert (one["bases"], one["checksum"]) == _model(1, 2)
    assert (two["bases"], two["checksum"]) == _model(2, 2)
    assert two["bases"] == 2 * one["bases"]  # weak scaling: per-GPU work is fixed
