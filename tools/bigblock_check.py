#!/usr/bin/env python3
"""One-off scale check (GPU box): a 500 Mbp target block (BASELINE configs[2] block size) against the oracle on a few
250 kbp chunks, both strands -- exercises >4 G-entry offsets, 60+ M hits per call, multi-batch extension and the
candidate-list growth paths.  usage: python tools/bigblock_check.py [target_mbp] [chunks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from helpers import Case, seg_equal  # noqa: E402
from oracle import oracle as O  # noqa: E402
from segalign_amd import engine as E, synth  # noqa: E402

tmbp = float(sys.argv[1]) if len(sys.argv) > 1 else 500.0
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
tlen = int(tmbp * 1e6)
target = synth.random_dna(tlen, 5)
target = synth.soft_mask(target, 6, 0.3, 200, 2000)
target = synth.join_records([target[i:i + tlen // 4] for i in range(0, tlen, tlen // 4)][:4])
# query block: 2 Mbp made of diverged / inverted pieces of distant target regions
rng = np.random.default_rng(7)
pieces = []
for i in range(8):
    p = int(rng.integers(0, target.size - 300000))
    seg = synth.mutate(target[p:p + 250000].copy(), 100 + i, 0.02 + 0.01 * i, indel_every=900)
    pieces.append(synth.reverse_complement(seg) if i % 3 == 0 else seg)
query = np.concatenate(pieces)
print("generated in %.1f s: target %d, query %d" % (time.time() - t0, target.size, query.size), flush=True)
t0 = time.time()
c = Case(target, query).oracle_setup(O)
print("oracle setup %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
c.engine_setup(E)
print("engine setup %.2f s (filter mode %d, lookup mode %d, %.1f M neighbourhood entries = %.1f GB of context records)" %
      (time.time() - t0, E.filter_mode(), E.lookup_mode(), E.neighbourhood_entries() / 1e6, E.neighbourhood_entries() * 32 / 1e9), flush=True)
assert np.array_equal(E.copy_index_table(), c.o_index)
ok = True
for rev in (False, True):
    for (s, e) in c.chunks()[:nchunks]:
        t1 = time.time()
        got = E.SeedAndFilterRange(s, e, rev, 0)
        t2 = time.time()
        st = E.last_call_stats()
        seeds = c.host_seeds(s, e, rev)
        want, ost = c.oracle_saf(seeds, rev)
        same = seg_equal(got, want)
        ok &= same
        print("rev=%d chunk %d-%d: hits %d candidates %d anchors %d  gpu %.1f ms  oracle %.1f s  %s" %
              (rev, s, e, st["num_hits"], st["num_candidates"], got.size - 1, (t2 - t1) * 1e3, time.time() - t2,
               "OK" if same else "MISMATCH"), flush=True)
E.ShutdownProcessor()
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
