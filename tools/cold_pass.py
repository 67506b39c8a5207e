#!/usr/bin/env python3
"""What does the FIRST pass over a query block cost after engine start (a one-block genome pair is exactly one such pass)?
usage (GPU box): python tools/cold_pass.py [one-chunk-call-first: 0/1]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from segalign_amd import engine as E, shard, synth  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
xdrop, hspthresh, seed_size = 910, 3000, 19
E.select_devices([0])
E.InitializeInterface(1)
kmer = E.GenerateShapePos(bench.SHAPE)
E.InitializeProcessor(True, 250000, seed_size, bench.default_sub_mat(xdrop), xdrop, hspthresh, False)
t, q = synth.make_pair(100_000_000, 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, indel_every=0, invert_frac=0.3, invert_block=100_000)
keep = E.SendRefWriteRequest(t, 0, t.size)
t0 = time.time()
E.GenerateSeedPosTable(keep, 0, t.size, 1, seed_size, kmer)
print("table %.1f ms" % ((time.time() - t0) * 1e3))
E.SendQueryWriteRequest(q, 0, q.size, 0)
jobs = shard.call_jobs(shard.plan_intervals(q.size, seed_size, 10_000_000), q.size - seed_size, 250000, E.lib().sa_get_chunks_per_call())
calls = [(j["a"], j["b"], j["rev"]) for j in jobs]
if first:
    t0 = time.time()
    E.SeedCalls([(0, 250000, False)], 0, 1)
    print("one-chunk call first: %.1f ms" % ((time.time() - t0) * 1e3))
for k in range(3):
    t0 = time.time()
    E.SeedCalls(calls, 0, 6)
    print("pass %d: %.1f ms" % (k + 1, (time.time() - t0) * 1e3))
E.ShutdownProcessor()
