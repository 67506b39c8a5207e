#!/usr/bin/env python3
"""A/B diagnostics of the extension filter on the bench workload (run on the GPU box):
   per-call candidate / survivor counts and single-stream kernel times for a set of env configurations.
   usage: python tools/filter_ab.py [target_mbp] -- each configuration re-runs InitializeProcessor in this process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from segalign_amd import engine as E, synth  # noqa: E402
from bench import SHAPE, default_sub_mat  # noqa: E402

tmbp = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
tlen = int(tmbp * 1e6)
target, query = synth.make_pair(tlen, 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, indel_every=0, invert_frac=0.3,
                                invert_block=100_000)
xdrop, hspthresh = 910, 3000
sub_mat = default_sub_mat(xdrop)
E.select_devices([0])
E.InitializeInterface(1)
kmer = E.GenerateShapePos(SHAPE)
CONFIGS = [dict(), dict(SEGALIGN_AMD_NO_PACKED_FILTER="1")] + [json.loads(a) for a in sys.argv[2:]]
ref = None
for cfg in CONFIGS:
    for k in list(os.environ):
        if k.startswith("SEGALIGN_AMD_"):
            del os.environ[k]
    os.environ.update({k: str(v) for k, v in cfg.items()})
    E.InitializeProcessor(True, 250000, len(SHAPE), sub_mat, xdrop, hspthresh, False)
    if ref is None:
        keep = E.SendRefWriteRequest(target, 0, target.size)
        E.GenerateSeedPosTable(keep, 0, target.size, 1, len(SHAPE), kmer)
        E.SendQueryWriteRequest(query, 0, query.size, 0)
        ref = True
    chunks = [(i * 250000, (i + 1) * 250000, rev) for i in range(8) for rev in (False, True)]
    for (a, b, rev) in chunks[:2]:
        E.SeedAndFilterRange(a, b, rev, 0)  # warm-up
    E.profile_reset()
    E.profile_enable(True)
    tot = dict(num_hits=0, num_candidates=0, num_survivors=0, num_anchors=0)
    sig = 0
    for (a, b, rev) in chunks:
        out = E.SeedAndFilterRange(a, b, rev, 0)
        st = E.last_call_stats()
        for k in tot:
            tot[k] += st[k]
        sig ^= hash(out.tobytes())
    E.profile_enable(False)
    prof = {name: round(ms * 1000.0 / max(n, 1), 1) for name, (ms, n) in E.profile_entries().items()}
    n = len(chunks)
    print(json.dumps(dict(cfg=cfg, per_call={k: v // n for k, v in tot.items()}, us=prof, sig=sig & 0xffffffff)))
