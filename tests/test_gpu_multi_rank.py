"""GPU: the N-rank path of bench.py with the REAL engine.  The driver launches `bench.py --gpus N` with one rank per GPU over RCCL;
a one-GPU box cannot do that, but everything except the transport can be shown on it: `--backend gloo --share-gpu` runs N ranks --
N processes, N engines, N copies of target + tables -- on HIP device 0, deals the calls of ONE pass to them (round-robin by default, like the reference's
dynamic pool; `--partition hits`: by seed hits, counted by a lookup-only pass every rank runs INSIDE the timed region) and reduces
time, bases, HSP count and the order-independent HSP checksum over gloo.
Checked: the 2-rank line reproduces the 1-rank checksum, HSP count and bases; and for the plumbing case (BASELINE configs[0]) the
checksum equals the one computed from the ORACLE's HSPs (src/seeder.cpp:47-121 + src/seed_filter.cu:682-828), so the number a
multi-GPU run is verified by is itself pinned to the CPU restatement."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from helpers import Case
from segalign_amd import shard

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nproc, extra, port, backend="gloo"):
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
            "--backend", backend, "--share-gpu"] + extra
    if nproc == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SEGALIGN_AMD_ARENA_GB="8")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def same_pass(a, b):
    ca, cb = a["config"], b["config"]
    assert ca["hsp_checksum"] == cb["hsp_checksum"] and ca["hsp_checksum"] != 0
    assert ca["hsps_per_step"] == cb["hsps_per_step"] and ca["hsps_per_step"] > 0
    assert ca["query_bases_per_step"] == cb["query_bases_per_step"]
    assert ca["calls_per_step"] == cb["calls_per_step"]


def test_two_ranks_reproduce_one_rank_on_the_plumbing_case_and_the_oracle(oracle):
    one = run_bench(1, ["--workload", "plumbing", "--partition", "hits"], 29611)
    two = run_bench(2, ["--workload", "plumbing"], 29612)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    # the default map is round-robin (the reference has no weighting pass): nothing is counted, nothing is hidden outside the timing
    assert "gloo" in two["config"]["backend"] and two["config"]["partition_cost_ms"] is None and two["config"]["partition"] == "round-robin"
    # --partition hits: the lookup-only pass runs once per pass INSIDE the timed region
    assert one["config"]["partition"] == "hits" and 0 < one["config"]["partition_cost_ms"] < one["ms_per_step"]
    # the call grain does not depend on the number of ranks
    assert one["config"]["chunks_per_call"] == two["config"]["chunks_per_call"]
    same_pass(one, two)
    # the same pass through the oracle: every 250 kbp chunk of [0, len - 19) on both strands (seeder.cpp:47-121)
    import bench
    wl = bench.make_workload(types.SimpleNamespace(workload="plumbing", target_fasta=None, query_fasta=None, target_mbp=None))
    c = Case(wl["target"], wl["query"], chunk=250_000).oracle_setup(oracle)
    chk, n = 0, 0
    for rev in (False, True):
        for (s, e) in c.chunks():
            segs, _ = c.oracle_saf(c.host_seeds(s, e, rev), rev)
            chk = (chk + shard.hsp_checksum(segs[1:], rev)) % bench.CHECK_MOD
            n += segs.size - 1
    assert n == one["config"]["hsps_per_step"] and n > 0
    assert chk == one["config"]["hsp_checksum"]


def test_two_ranks_reproduce_one_rank_on_a_20_mbp_stand_in():
    args = ["--workload", "ce11cb4", "--target-mbp", "20", "--chunks-per-call", "10"]
    one = run_bench(1, args, 29613)
    two = run_bench(2, args + ["--partition", "hits"], 29614)
    same_pass(one, two)
    assert one["config"]["calls_per_step"] == 16   # 2 intervals x 2 strands x 40 chunks in calls of 10
    assert one["config"]["partition_cost_ms"] is None
    assert 0 < two["config"]["partition_cost_ms"] < two["ms_per_step"]   # the weighting pass is timed
    imb = two["config"]["partition_imbalance"]
    assert imb["ranks"] == 2 and 1.0 <= imb["by_hits"] < 1.2 and 1.0 <= imb["round_robin"] < 1.5


def test_a_failing_rccl_group_does_not_take_the_run_down():
    """`--backend nccl` (the driver's launch) with both ranks on ONE device: RCCL refuses ("Duplicate GPU detected").  The transport
    only carries a barrier and three tiny reductions, so bench.py keeps its gloo default group, says in its line that the reductions
    went over gloo and why -- and the pass is the same pass."""
    one = run_bench(1, ["--workload", "plumbing", "--partition", "hits"], 29615)
    two = run_bench(2, ["--workload", "plumbing"], 29616, backend="nccl")
    assert two["n_gpus"] == 2 and two["config"]["backend"].startswith("gloo (nccl failed")
    same_pass(one, two)
