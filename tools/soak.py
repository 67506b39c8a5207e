#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random block pairs (uniform + repeat-rich), seeds (12of19 / 14of22), steps, scoring
parameters, chunk sizes and call shapes -- every HSP vector of the engine against the oracle (checker only), through the drop-in
entry (host seed words), the device seeder, multi-chunk calls and the interval entry (table-direct calls of several chunks, both
strands, 1-3 host threads).  Runs until SOAK_SECONDS (default 300) are used up; prints one line per case and a summary; exits
non-zero at the first mismatch (the failing case's seed is on its line).   python tools/soak.py [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import Case, seg_equal  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)
from segalign_amd import engine as E, shard, synth  # noqa: E402

def shape_14of22():
    try:
        import re
        src = open(os.path.join(ROOT, "tests", "test_gpu_edge_cases.py")).read()
        m = re.search(r'"([T0]{22})"', src)
        return m.group(1) if m else None
    except Exception:
        return None


def build_pair(rng, seed):
    kind = rng.choice(["uniform", "uniform", "repeats", "collinear"])
    n = int(rng.integers(30000, 400000))
    if kind == "collinear":
        t, q = synth.make_pair(n, seed, seed + 1, sub_rate=float(rng.uniform(0.01, 0.06)), invert_frac=0.0,
                               indel_every=int(rng.choice([0, 0, 300, 2000])))
    else:
        t, q = synth.make_pair(n, seed, seed + 1, sub_rate=float(rng.uniform(0.03, 0.2)), mask_frac=float(rng.uniform(0, 0.2)),
                               records=int(rng.integers(1, 4)), indel_every=int(rng.choice([0, 150, 600])),
                               n_runs=int(rng.integers(0, 3)), invert_frac=float(rng.uniform(0, 0.5)),
                               invert_block=int(rng.integers(2000, 20000)))
    if kind == "repeats":  # a dispersed repeat family + microsatellites: heavy buckets, crowded diagonals
        fam = synth.random_dna(int(rng.integers(300, 1500)), seed + 7)
        for arr in (t, q):
            for _ in range(int(rng.integers(5, 60))):
                p = int(rng.integers(0, arr.size - fam.size - 1))
                cp = synth.mutate(fam, int(rng.integers(1 << 30)), float(rng.uniform(0.0, 0.15)), 0)
                arr[p:p + cp.size] = cp[:arr.size - p]
            for _ in range(int(rng.integers(0, 4))):
                unit = synth.random_dna(int(rng.integers(1, 5)), int(rng.integers(1 << 30)))
                ln = int(rng.integers(50, 600))
                p = int(rng.integers(0, arr.size - ln - 1))
                arr[p:p + ln] = np.tile(unit, ln // unit.size + 1)[:ln]
    return kind, t, q


def one_case(seed, s14):
    rng = np.random.default_rng(seed)
    kind, t, q = build_pair(rng, seed)
    shape = s14 if (s14 and rng.random() < 0.1) else None  # (src/main.cpp:164-167; the string is read from tests/test_gpu_edge_cases.py)
    kw = dict(step=int(rng.choice([1, 1, 1, 2, 5])), transition=bool(rng.random() < 0.7),
              xdrop=int(rng.choice([300, 910, 910, 2500])), hspthresh=int(rng.choice([1500, 3000, 3000, 5000])),
              noentropy=bool(rng.random() < 0.3), chunk=int(rng.choice([8000, 25000, 60000, 250000])))
    if shape:
        kw["shape"] = shape
    E.reset_option(None)
    ko = bool(rng.random() < 0.3)   # the multi-chunk calls of (2) key-ordered (option key_order = 2: join.hip, extend.hip 1e)
    if ko:
        E.set_option("key_order", 2)
    audit = bool(rng.random() < 0.3)  # every hit the filter levels REJECT in the per-chunk calls of (1) extended by the oracle: none may pass
    if audit:
        E.set_option("audit_cap", 1 << 22)
    if rng.random() < 0.15:
        E.set_option("ctx_skip_seed", 0)  # (the context layout with the seed window inside the record)
    if rng.random() < 0.2:
        E.set_option("l2_right_state", 1)  # (the second level resumes an open right walk behind the class filter's context)
    # MAX_HITS of a smaller GPU (src/seed_filter.cu:832-841, hazard H4): chunks at or above it run in the reference's greedy iteration
    # groups -- planned table-direct up to eight per chunk (probe.hip), by the general path beyond
    mh = int(rng.choice([1 << 30, 1 << 30, 1 << 30, 400, 3000, 20000, 150000]))
    c = Case(t, q, **kw).oracle_setup(O).engine_setup(E)
    if mh < (1 << 30):  # (the engine's general path plans at most 1000 iterations per chunk: keep the heaviest chunk below ~500)
        per = E.CountCallHits([(s_, e_, r_) for r_ in (False, True) for (s_, e_) in c.chunks()], 0, 2)
        mh = max(mh, max(per + [0]) // 500 + 1)
    E.set_max_hits(mh if mh < (1 << 30) else 0)
    kw["max_hits"] = mh
    kw["key_order"] = ko
    kw["audit"] = audit
    hsps = hits = 0
    audited = [0]
    try:
        q_len = c.query.size - c.seed_size
        # (1) drop-in entry + device seeder, chunk by chunk
        want = {False: [], True: []}
        for rev in (False, True):
            for (s, e) in c.chunks():
                seeds = c.host_seeds(s, e, rev)
                if seeds.size == 0:
                    want[rev].append(None)
                    continue
                w, st = c.oracle_saf(seeds, rev, max_hits=mh)
                want[rev].append(w)
                hits += st["num_hits"]
                hsps += w.size - 1
                # (the reference asserts num_seeds <= 13 * chunk, seed_filter.cu:688-692: 14of22 with transitions exceeds it)
                if seeds.size <= (13 if c.transition else 1) * c.chunk:
                    assert seg_equal(E.SeedAndFilter(seeds, rev, 0), w), ("drop-in", rev, s, e)
                assert seg_equal(E.SeedAndFilterRange(s, e, rev, 0), w), ("range", rev, s, e)
                if audit and st["num_hits"] > 0 and E.last_call_stats()["lookup_path"] != 0:
                    pairs, n = E.get_audit()
                    assert n == pairs.shape[0], "audit list overflowed"
                    if n:
                        ok, _ = O.extend_hits_pass(c.o_ref, c.o_qrc if rev else c.o_q, c.sub_mat, pairs, xdrop=c.xdrop, hspthresh=c.hspthresh,
                                                   noentropy=c.noentropy)
                        assert not ok.any(), ("a filter level rejected a passing hit", rev, s, e, pairs[np.nonzero(ok)[0][:3]].tolist())
                        audited[0] += int(n)
        # (2) the interval entry over the whole block (multi-chunk table-direct calls), against the same oracle vectors
        for threads in (int(rng.integers(1, 4)),):
            fw, rc, tot = E.SeedInterval(0, q_len, q_len, E.STRAND_BOTH, 0, threads)
            for rev, got in ((False, fw), (True, rc)):
                exp = []
                for (s, e) in shard.chunks_of((0, q_len), c.chunk, q_len, rev):
                    seeds = c.host_seeds(s, e, rev)
                    if seeds.size:
                        exp.append(c.oracle_saf(seeds, rev, max_hits=mh)[0][1:])
                exp = np.concatenate(exp) if exp else np.zeros(0, dtype=fw.dtype)
                assert seg_equal(got, exp), ("interval", rev, threads, got.size, exp.size)
    finally:
        E.set_max_hits(0)
        E.ShutdownProcessor()
    kw["audited"] = audited[0]
    return kind, kw, t.size, hits, hsps


def main():
    O.build(with_ref=False)
    budget = float(os.environ.get("SOAK_SECONDS", "300"))
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    s14 = shape_14of22()
    t0 = time.time()
    n = 0
    tot_hits = tot_hsps = 0
    while time.time() - t0 < budget:
        try:
            kind, kw, size, hits, hsps = one_case(seed, s14)
        except AssertionError as ex:
            print("MISMATCH seed %d: %s" % (seed, ex), flush=True)
            sys.exit(1)
        print("seed %d %-9s %7d bp step %d %s chunk %6d xdrop %4d thresh %4d %s%s: %d hits, %d HSPs ok" % (
            seed, kind, size, kw["step"], "tr" if kw["transition"] else "no-tr", kw["chunk"], kw["xdrop"], kw["hspthresh"],
            "noentropy " if kw["noentropy"] else "", ("14of22" if "shape" in kw else "12of19") + (" key-ordered" if kw.get("key_order") else "") + (" audited %d" % kw["audited"] if kw.get("audit") else "") + (" MAX_HITS %d" % kw["max_hits"] if kw["max_hits"] < (1 << 30) else ""), hits, hsps), flush=True)
        n += 1
        tot_hits += hits
        tot_hsps += hsps
        seed += 1
    print("soak: %d cases, %d seed hits, %d HSPs, all bit-identical to the oracle, %.0f s" % (n, tot_hits, tot_hsps, time.time() - t0))


if __name__ == "__main__":
    main()
