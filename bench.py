#!/usr/bin/env python3
"""bench.py -- Gbp of query seeded + filtered + extended per second on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  One JSON line on stdout from rank 0.

Default workload = BASELINE.json configs[1] (ce11 x cb4, default 12of19 seed, 1 x MI355X).  The real assemblies cannot be
downloaded here, so the stand-in of BASELINE.md / SURVEY.md 8(d) is generated deterministically (a ~100 Mbp, 7-record
target and a query that is an 8 %-diverged copy with 20 % soft-masked runs and inversions); real FASTA files are taken
with --target-fasta / --query-fasta.  A *step* is ONE PASS OVER THE WHOLE QUERY BLOCK: every 10 Mbp interval (the
reference's lastz_interval, src/graph.h:11) on BOTH strands against the resident target -- per interval 40 + 40
SeedAndFilter chunk calls of 250 kbp (DEFAULT_WGA_CHUNK), seeds generated on the device so that the whole
"seeded + filtered + extended" metric is inside the timed region and no seed vector crosses PCIe.  Target upload, encoding
and the seed-table builds happen before the timed region (reported separately, as the reference does under --debug,
src/main.cpp:617-629).

Other workloads (--workload): `notransition` = configs[4] (--notransition --step=1), `rm` = configs[3] (repeat-masker
path: the target self-aligned through sa_rm_mask_interval over the reference's interval plan), `human` = one block pair
of configs[2] (a 500 Mbp target block, the size at which the reference closes a block, x a 100 Mbp query block of 1.2
%-diverged shuffled pieces; every rank of an N-GPU run holds its own block pair, as the 6 x 6 block pairs of a 3 Gbp x 3 Gbp
run are independent), `plumbing` = configs[0] (1 Mbp x 1 Mbp).

Multi-GPU: query intervals are independent shards: in every step rank r walks the interval list starting at
interval r (r, r+1, ... wrapping), so per-GPU work is fixed => "scaling": "weak"; every rank holds target + tables; there
is NO data-path collective (torch.distributed only carries the barrier and the max/sum of the timing).
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = "TTT0T00TT00T0T0TTTT"  # 12of19, src/main.cpp:160-163
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RANDOM_LINES_PER_S = 57e9      # measured on MI355X: random 128-byte line gathers per second (tools/micro/gather_bw.hip)
FILTER_KERNELS = {0: "extend_filter_kernel", 1: "extend_filter_kernel", 3: "extend_filter_packed_kernel"}
# event scope of the engine (sa_profile_*) -> device kernels launched inside it (names as rocprofv3 prints them)
SCOPE_KERNELS = {
    "seed_probe": ["probe_kernel"], "probe_compact": ["probe_partials_kernel", "probe_compact_kernel"],
    "iteration_plan": ["probe_plan_kernel", "plan_kernel"], "seed_lookup": ["seed_lookup_kernel"],
    "expand_hits": ["expand_hits_kernel"], "extend_filter": ["extend_filter_packed_kernel"],  # (the default, packed filter)
    "chain_group": ["chain_count_kernel", "chain_scan_kernel", "chain_scatter_kernel", "chain_bucket_sort_kernel"],
    "chain_link": ["chain_link_kernel"], "extend_exact_chain": ["extend_exact_chain_kernel"],
    "extend_exact": ["extend_exact_kernel"], "extend_entropy": ["extend_entropy_kernel"], "dedup_seg": ["dedup_seg_kernel"],
}
# context-table calls (lookup mode 2): the filter is two kernels -- level 1 on the context records, level 2 (the packed kernel)
# on the few hits level 1 could not decide
CTX_SCOPE_KERNELS = {"extend_filter": ["extend_filter_ctx_kernel"], "extend_filter2": ["extend_filter_packed_kernel"]}
EXTENSION_SCOPES = ["extend_filter", "extend_filter2", "chain_group", "chain_link", "extend_exact_chain", "extend_exact", "extend_entropy"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ce11cb4", choices=["ce11cb4", "notransition", "rm", "human", "plumbing"])
    ap.add_argument("--target-fasta", default=None, help="real target FASTA (e.g. ce11.fa[.gz]); first 500 Mbp block is used")
    ap.add_argument("--query-fasta", default=None, help="real query FASTA (e.g. cb4.fa[.gz]); first 500 Mbp block is used")
    ap.add_argument("--target-mbp", type=float, default=None, help="synthetic target size (default per workload)")
    ap.add_argument("--interval", type=int, default=10_000_000)
    ap.add_argument("--chunk", type=int, default=250_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU baseline budget")
    ap.add_argument("--intervals-in-flight", type=int, default=2,
                    help="query intervals processed concurrently (each with --host-threads calls in flight), like the reference's seeder threads")
    ap.add_argument("--one-interval", action="store_true",
                    help="profiling aid: set up, run ONE interval with one call in flight, exit (short enough for --pmc passes)")
    ap.add_argument("--host-threads", type=int, default=2,
                    help="host threads issuing SeedAndFilter calls (the reference runs one TBB seeder body per core; "
                         "the engine has 2 slots per device so one call's syncs overlap another call's kernels)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-kernel HIP events in the timed region (roofline block from the untimed passes only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only rehearsal of the launch/shard/reduce/JSON contract (gloo, no GPU): real shard + chunk "
                         "arithmetic around a stub engine")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def make_workload(args, rank=0):
    """-> dict(target, query, transition, rm, label, data).  Deterministic; rank only matters for `human` (own block pair)."""
    from segalign_amd import synth
    w = args.workload
    if args.target_fasta or args.query_fasta:
        from segalign_amd import fasta
        if not (args.target_fasta and (args.query_fasta or w == "rm")):
            raise SystemExit("--target-fasta and --query-fasta must be given together (rm: target only)")
        target, tb = fasta.first_block(args.target_fasta)
        query, qb = (target, tb) if w == "rm" else fasta.first_block(args.query_fasta)
        return dict(target=target, query=query, transition=w != "notransition", rm=w == "rm", data="fasta",
                    label="%s x %s (first block of %d / %d), %s" % (os.path.basename(args.target_fasta),
                                                                    os.path.basename(args.query_fasta or args.target_fasta), tb, qb, w))
    if w in ("ce11cb4", "notransition", "rm"):
        tlen = int((args.target_mbp or 100.0) * 1e6)
        target, query = synth.make_pair(tlen, 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, indel_every=0,
                                        invert_frac=0.3, invert_block=100_000)
        label = {"ce11cb4": "ce11 x cb4 stand-in (BASELINE configs[1]): %.0f Mbp 7-record target x 8%%-diverged soft-masked query, "
                            "12of19 + transitions",
                 "notransition": "ce11 x cb4 stand-in with --notransition --step=1 (BASELINE configs[4]): %.0f Mbp 7-record target x "
                                 "8%%-diverged soft-masked query, 12of19, one seed word per position",
                 "rm": "repeat-masker path (BASELINE configs[3]): %.0f Mbp ce11 stand-in self-aligned, neighbor_proportion 0.2, M 1"}[w]
        return dict(target=target, query=target if w == "rm" else query, transition=w != "notransition", rm=w == "rm",
                    data="synthetic", label=label % (tlen / 1e6))
    if w == "plumbing":
        target, query = synth.make_pair(int((args.target_mbp or 1.0) * 1e6), 1, 2, sub_rate=0.15, indel_every=500, invert_frac=0.0)
        return dict(target=target, query=query, transition=True, rm=False, data="synthetic",
                    label="plumbing case (BASELINE configs[0]): 1 Mbp uniform target x 15%-substituted copy with sparse indels")
    # human: one 500 Mbp target block (4 records) of a 24 x 125 Mbp genome, and a 100 Mbp query block made of 1-10 Mbp pieces
    # of the same block, 1.2 % diverged, shuffled, every third piece inverted; own pair per rank
    tlen = int((args.target_mbp or 500.0) * 1e6)
    t = synth.random_dna(tlen, 5 + 100 * rank)
    t = synth.soft_mask(t, 6 + 100 * rank, 0.3, 200, 2000)
    per = tlen // 4
    target = synth.join_records([t[i * per:(i + 1) * per] for i in range(4)])
    del t
    rng = np.random.default_rng(7 + rank)
    pieces, total, i = [], 0, 0
    qlen = min(100_000_000, tlen // 2)
    while total < qlen:
        n = int(rng.integers(1_000_000, 10_000_001))
        n = min(n, qlen - total)
        p = int(rng.integers(0, target.size - n))
        seg = synth.mutate(target[p:p + n], 1000 + i + 100 * rank, 0.012)
        pieces.append(synth.reverse_complement(seg) if i % 3 == 0 else seg)
        total += n
        i += 1
    return dict(target=target, query=np.concatenate(pieces), transition=True, rm=False, data="synthetic",
                label="human-scale block pair (BASELINE configs[2]): %.0f Mbp 4-record target block x %.0f Mbp query block of "
                      "1.2%%-diverged shuffled 1-10 Mbp pieces, 12of19 + transitions" % (tlen / 1e6, total / 1e6))


def my_intervals(items, rank, step):
    """Weak scaling: every rank walks ALL items each step, rank r starting at item r (+ the step number)."""
    n = len(items)
    return [items[(rank + step + i) % n] for i in range(n)]


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import torch
    from segalign_amd import shard
    dist = None
    dev = "cpu" if args.dry_run else "cuda"
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.dry_run:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # "nccl" IS RCCL on ROCm
    elif not args.dry_run:
        torch.cuda.set_device(local_rank)

    if args.dry_run:
        return dry_run(args, rank, world, dist, torch, shard)

    from segalign_amd import engine as E

    t_gen0 = time.time()
    wl = make_workload(args, rank)
    target, query = wl["target"], wl["query"]
    t_gen = time.time() - t_gen0

    # default parameters of the reference (src/main.cpp:61-124)
    xdrop, hspthresh, seed_size = 910, 3000, len(SHAPE)
    sub_mat = default_sub_mat(xdrop)

    E.select_devices([local_rank])
    E.InitializeInterface(1)
    kmer = E.GenerateShapePos(SHAPE)
    # one engine slot per call in flight (the engine's default is 2)
    os.environ.setdefault("SEGALIGN_AMD_SLOTS", str(max(2, max(1, args.host_threads) * max(1, args.intervals_in_flight))))
    E.InitializeProcessor(wl["transition"], args.chunk, seed_size, sub_mat, xdrop, hspthresh, False)
    t0 = time.time()
    keep = E.SendRefWriteRequest(target, 0, target.size)
    t_ref = time.time() - t0
    t0 = time.time()
    E.GenerateSeedPosTable(keep, 0, target.size, 1, seed_size, kmer)
    t_table = time.time() - t0
    t0 = time.time()
    if wl["rm"]:
        E.RmSendQueryWriteRequest()
    else:
        E.SendQueryWriteRequest(query, 0, query.size, 0)
    t_query = time.time() - t0

    if wl["rm"]:
        items = [(t["start"], t["end"], t["ref_start"], t["ref_end"]) for t in shard.rm_plan(target.size, seed_size=seed_size,
                                                                                             lastz_interval_size=args.interval)]
    else:
        items = shard.plan_intervals(query.size, seed_size, args.interval)  # src/main.cpp:383-393 over [0, len - seed_size)
    q_block_len = query.size - seed_size  # q_len handed to the seeder (main.cpp:708)

    def run_item(it, collect=None, threads=None):
        """one interval: seeder_body::operator() (src/seeder.cpp:12-127; repeat_masker_src/seeder.cpp:28-195 for rm) -- every
        250 kbp chunk of both strands through the engine, chunk calls issued by the library's own C++ threads"""
        if wl["rm"]:
            iv, tot = E.RmMaskInterval(it[0], it[1], it[2], it[3], E.STRAND_BOTH, 1)
            if collect is not None:
                collect.append(dict(num_seeds=tot["num_seeds"], num_hits=tot["num_hits"], num_survivors=tot["num_hsps"],
                                    num_candidates=0, num_examined=0, num_examined_filter=0))
            return it[1] - it[0], int(iv.size)
        fw, rc, st = E.SeedInterval(it[0], it[1], q_block_len, E.STRAND_BOTH, 0, max(1, threads or args.host_threads))
        if collect is not None:
            collect.append(st)
        return it[1] - it[0], int(fw.size + rc.size)

    def run_step(k, collect=None, threads=None):
        todo = my_intervals(items, rank, k)
        nt = max(1, threads or args.host_threads)
        if wl["rm"] and nt > 1:
            # sa_rm_mask_interval walks the chunks of ONE interval on one engine slot; intervals are independent
            # (repeat_masker_src/main.cpp hands them to parallel seeder bodies): keep `nt` of them in flight
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nt) as pool:
                res = list(pool.map(lambda it: run_item(it, collect, threads), todo))
            return sum(r[0] for r in res), sum(r[1] for r in res)
        if args.intervals_in_flight > 1 and threads is None and len(todo) > 1:
            # the reference host keeps several seeder bodies (intervals) in flight (src/main.cpp:601-737, one per TBB thread):
            # the calls of one interval finish together, a second interval fills the gap
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(args.intervals_in_flight) as pool:
                res = list(pool.map(lambda it: run_item(it, collect, threads), todo))
            return sum(r[0] for r in res), sum(r[1] for r in res)
        b = h = 0
        for it in todo:
            bb, hh = run_item(it, collect, threads)
            b += bb
            h += hh
        return b, h

    if args.one_interval:
        nb, nh = run_item(items[0], None, 1)
        print(json.dumps({"one_interval": True, "bases": nb, "hsps": nh, "workload": args.workload}))
        E.ShutdownProcessor()
        return

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- warmup (untimed) ----------------
    for w in range(args.warmup):
        run_step(w)

    # ---------------- timed region ----------------
    E.profile_reset()
    E.profile_enable(not args.no_kernel_events)
    call_stats = []
    barrier()
    t0 = time.perf_counter()
    bases = hsps = 0
    for k in range(args.steps):
        b, h = run_step(k, call_stats)
        bases += b
        hsps += h
    barrier()
    elapsed = time.perf_counter() - t0
    E.profile_enable(False)
    prof = E.profile_entries()

    # max over ranks, sum of bases
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([bases, hsps], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        bases, hsps = int(tb[0].item()), int(tb[1].item())

    if rank == 0 and not prof and args.no_kernel_events:  # event-free timed region: kernel times from one extra (untimed) pass
        E.profile_reset()
        E.profile_enable(True)
        call_stats = []
        run_step(0, call_stats)
        E.profile_enable(False)
        prof = E.profile_entries()
    roof = None
    if rank == 0 and prof:
        roof = roofline(args, E, wl, prof, call_stats, run_step, run_item, items)

    # ---------------- CPU baseline (rank 0, N == 1 only, bounded sample) ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not wl["rm"]:
        cpu = cpu_baseline(E, target, query, sub_mat, seed_size, kmer, args, xdrop, hspthresh, wl["transition"])

    if rank == 0:
        value = bases / elapsed / 1e9
        line = {
            "metric": "Gbp query seeded+filtered+extended per sec", "value": round(value, 5), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": wl["data"],
            "config": {"workload": wl["label"] + ", HOXD70, xdrop 910, hspthresh 3000; step = one pass over the whole %d bp query "
                                                 "block (%d intervals of %d bp), both strands, %d bp chunks (%d chunks of a strand share one "
                                                 "pass over the kernels), device-side seeding" % (query.size, len(items), args.interval, args.chunk,
                                                                                                 E.lib().sa_get_chunks_per_call()),
                       "workload_key": args.workload,
                       "parallelism": "query-interval shards x%d (every rank walks the interval list from its own offset), no collective" % world,
                       "hsps_per_step": hsps // max(args.steps * world, 1)},
            "setup_s": {"generate": round(t_gen, 2), "target_upload_encode": round(t_ref, 3),
                        "seed_table_build": round(t_table, 3), "query_upload_encode": round(t_query, 3)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        # table build (setup, once per target block): algorithmic bytes of the reference-layout table, and the bytes
        # this build writes on top of it -- the neighbourhood table, 4-byte positions or 32-byte context records
        lm, nent = E.lookup_mode(), E.neighbourhood_entries()
        tv = float(target.size)
        alg_tb = tv + 8.0 * 4 ** 12 + 8.0 * tv
        impl_tb = alg_tb + nent * {0: 0, 1: 4, 2: 32}[lm]
        line["table_build"] = {
            "seconds": round(t_table, 4), "lookup_mode": lm, "neighbourhood_entries": nent,
            "bytes": "1*T + 8*4^12 + 8*T_valid (reference-layout table) [+ 32 B (context) or 4 B per neighbourhood entry]",
            "algorithmic_bytes": int(alg_tb), "algorithmic_frac": round(alg_tb / max(t_table, 1e-9) / 1e9 / HBM_PEAK_GBS, 5),
            "written_bytes": int(impl_tb), "written_frac": round(impl_tb / max(t_table, 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "includes hipMalloc of the tables (first block of a process) and every sync of the build"}
        print(json.dumps(line))
        sys.stdout.flush()
    E.ShutdownProcessor()
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# roofline block
# ------------------------------------------------------------------------------------------------------------------
def roofline(args, E, wl, prof, call_stats, run_step, run_item, items):
    """Per-kernel time from HIP events on the engine's own streams; algorithmic bytes per SURVEY 8(d) / DESIGN.md 4."""
    # the same kernels without a second call overlapping them: one extra (untimed) pass issued by ONE host thread
    E.profile_reset()
    E.profile_enable(True)
    solo_stats = []
    run_step(0, solo_stats, threads=1)
    E.profile_enable(False)
    solo = E.profile_entries()
    # per-hit ratios E/H and E_filter/H from one instrumented (untimed) interval: deterministic, identical work
    e_all = e_flt = 0.0
    if not wl["rm"]:
        E.set_count_examined(True)
        sample = []
        run_item(items[0], sample)
        E.set_count_examined(False)
        sH = max(sum(s["num_hits"] for s in sample), 1)
        e_all = sum(s["num_examined"] for s in sample) / sH          # E per hit (reference algorithm)
        e_flt = sum(s["num_examined_filter"] for s in sample) / sH   # bases per hit scored by the filter kernel

    def totals(stats):
        return (sum(s["num_hits"] for s in stats), sum(s["num_survivors"] for s in stats), sum(s["num_seeds"] for s in stats),
                sum(s["num_candidates"] for s in stats))

    H, A, S, Cn = totals(call_stats)
    sH, sA, sS, sC = totals(solo_stats)
    table_direct = "seed_probe" in prof
    ctx_filter = "extend_filter2" in prof   # context-table calls: filter = level 1 (context records) + level 2 (packed kernel)
    scope_kernels = dict(SCOPE_KERNELS, **(CTX_SCOPE_KERNELS if ctx_filter else {}))
    filter_scopes = ["extend_filter", "extend_filter2"] if ctx_filter else ["extend_filter"]
    # algorithmic bytes (restated per kernel in DESIGN.md 4):
    #   seed lookup                      : 16*S            (8 B seed word + 8 B bucket extent per seed word)
    #   lookup + expansion               : 16*S + 12*H     (hit list materialised)   /  16*S + 4*H  (fused into extension)
    #   X-drop filter                    : h*H + 2*E_filter + 12*C   with h = 8 (hit records in) or 4 (table-direct: run entries in)
    #   extension as a whole             : h*H + 2*E + 20*A
    hin = 4.0 if table_direct else 8.0

    def alg(h, a, s, c):
        return {"seed_lookup": 16.0 * s, "expand_hits": 12.0 * h, "extend_filter": hin * h + 2.0 * e_flt * h + 12.0 * c,
                "extension_total": hin * h + 2.0 * e_all * h + 20.0 * a,
                "lookup_expand": 16.0 * s + (4.0 if table_direct else 12.0) * h}

    alg_t, alg_s = alg(H, A, S, Cn), alg(sH, sA, sS, sC)
    lookup_scope = "seed_probe" if table_direct else "seed_lookup"

    def ms_of(p, scopes):
        return sum(p[k][0] for k in scopes if k in p)

    def rate(nbytes, ms):
        return (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else None

    def block(name, scopes, nbytes_t, nbytes_s, formula):
        a_t, a_s = rate(nbytes_t, ms_of(prof, scopes)), rate(nbytes_s, ms_of(solo, scopes))
        return {"scopes": scopes, "bytes": formula,
                "achieved": round(a_t, 1) if a_t else None, "frac": round(a_t / HBM_PEAK_GBS, 4) if a_t else None,
                "single_stream": {"achieved": round(a_s, 1) if a_s else None, "frac": round(a_s / HBM_PEAK_GBS, 4) if a_s else None}}

    kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1], "avg_us": round(1e3 * v[0] / max(v[1], 1), 2)}
               for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    gpu_ms = sum(v[0] for v in prof.values())
    name, (ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
    if name == "extend_filter" and ctx_filter:
        # 4*H + 2*E_filter + 12*C is the work of the WHOLE filter stage (every hit scored until it drops), so it is divided by
        # the time of both levels: level 1 alone only looks at the 48 + 64 context bases of a hit
        ms = ms_of(prof, filter_scopes)
    per_scope_bytes = {"extend_filter": ("extend_filter", "%g*H + 2*E_filter + 12*C" % hin), lookup_scope: ("seed_lookup", "16*S"),
                       "expand_hits": ("expand_hits", "12*H")}
    key, formula = per_scope_bytes.get(name, (None, None))
    achieved = rate(alg_t[key], ms) if key else None
    traffic_db, traffic_src, kstats = committed_profile(args, E)
    check = profile_check(prof, solo, kstats, args, scope_kernels)
    if name == "extend_filter":
        symbol = "extend_filter_ctx_kernel" if ctx_filter else FILTER_KERNELS.get(E.filter_mode())
    else:
        symbol = (scope_kernels.get(name) or [None])[0]

    def traffic_of(sym, launches_in_scope=1):
        if not traffic_db or not sym or sym not in traffic_db or not check["ok"]:
            return None
        return int(traffic_db[sym]["hbm_bytes"])

    def first_class(scope, alg_key, formula):
        """a kernel of its own standing: algorithmic fraction (timed region + single stream) and measured-traffic ratio"""
        if scope not in prof or not prof[scope][1]:
            return None
        sym = scope_kernels[scope][0]
        n_t, n_s = prof[scope][1], max(solo.get(scope, (0, 0))[1], 1)
        a_t = rate(alg_t[alg_key], prof[scope][0])
        a_s = rate(alg_s[alg_key], solo[scope][0]) if scope in solo else None
        tr = traffic_of(sym)
        abl = alg_t[alg_key] / n_t
        return {"kernel_symbol": sym, "bytes": formula, "algorithmic_bytes_per_launch": round(abl),
                "avg_launch_us": round(1e3 * prof[scope][0] / n_t, 2), "algorithmic_frac": round(a_t / HBM_PEAK_GBS, 4) if a_t else None,
                "single_stream": {"avg_launch_us": round(1e3 * solo[scope][0] / n_s, 2) if scope in solo else None,
                                  "algorithmic_frac": round(a_s / HBM_PEAK_GBS, 4) if a_s else None},
                "traffic": tr, "traffic_ratio": round(tr / abl, 3) if (tr and abl) else None}

    traffic = traffic_of(symbol)
    solo_ms = (ms_of(solo, filter_scopes) if (name == "extend_filter" and ctx_filter) else solo[name][0]) if name in solo else None
    s_avg_us = 1e3 * solo_ms / max(solo[name][1], 1) if name in solo else None
    single = None
    if key and name in solo and solo[name][1]:
        sg = rate(alg_s[key], solo_ms)
        single = {"avg_launch_us": round(s_avg_us, 2), "achieved": round(sg, 1), "frac": round(sg / HBM_PEAK_GBS, 4),
                  "note": "same kernel, one call in flight (no overlap with a second stream); untimed extra pass"}
    return {
        "bound": "hbm", "kernel": name, "kernel_symbol": symbol, "bytes": formula,
        "kernel_scopes": filter_scopes if name == "extend_filter" else [name],
        "note": ("context-table filter: the bytes are the algorithmic figure of the whole filter stage (1 byte per examined base and "
                 "sequence), the duration is level 1 + level 2; the kernel itself streams 32-byte context records + 4-bit query "
                 "windows (measured traffic below the algorithmic bytes) and is bound by VALU / LDS / address path, not HBM -- DESIGN.md 4"
                 ) if (name == "extend_filter" and ctx_filter) else None,
        "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
        "traffic": traffic, "traffic_source": traffic_src, "profile_check": check,
        "traffic_ratio": round(traffic / (alg_t[key] / max(launches, 1)), 3) if (traffic and key) else None,
        # the same traffic counted in 128-byte lines against the measured random-gather ceiling of the chip
        "random_line_roofline": ({"lines_per_launch": int(traffic // 128), "peak_lines_per_s": RANDOM_LINES_PER_S,
                                  "frac_single_stream": round(traffic / 128 / (s_avg_us * 1e-6) / RANDOM_LINES_PER_S, 4)}
                                 if (traffic and s_avg_us) else None),
        "calls_in_flight": max(1, args.host_threads) * max(1, args.intervals_in_flight), "single_stream": single,
        "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
        "algorithmic_bytes_per_launch": round(alg_t[key] / max(launches, 1)) if key else None,
        "dominant_share_of_gpu_time": round(ms / gpu_ms, 4) if gpu_ms else None,
        "table_direct": table_direct,
        "per_hit": {"examined_bases_E": round(e_all, 2), "examined_by_filter": round(e_flt, 2),
                    "candidate_frac": round(Cn / max(H, 1), 5), "survivor_frac": round(A / max(H, 1), 5),
                    "hits_per_seed_word": round(H / max(S, 1), 3)},
        # the kernel north_star names: seed lookup.  Table-direct: probe_kernel (one probe per query POSITION into the
        # neighbourhood table; the 13 seed words of a position are never materialised, so its traffic is BELOW 16*S)
        "seed_lookup": first_class(lookup_scope, "seed_lookup", "16*S"),
        "expand_hits": first_class("expand_hits", "expand_hits", "12*H"),
        "lookup_expand": block("lookup_expand", [lookup_scope, "probe_compact", "hit_prefix_scan", "expand_hits"], alg_t["lookup_expand"],
                               alg_s["lookup_expand"], "16*S + 4*H over probe + compact (run entries are read by the filter)"
                               if table_direct else "16*S + 12*H over lookup + prefix scan + expansion"),
        "extension_total": block("extension_total", EXTENSION_SCOPES, alg_t["extension_total"], alg_s["extension_total"],
                                 "%g*H + 2*E + 20*A over filter + chain grouping/link + exact + entropy kernels" % hin),
        "kernels": kernels,
    }


def committed_profile(args, E):
    """(traffic per kernel symbol, source path, kernel_stats averages) of the newest profiles/rNN collected on THIS workload
    shape; (None, None, None) if there is none.  bench.py cannot run the PMC passes on itself (separate rocprofv3 runs,
    tools/profile_bench.sh), so measured HBM traffic comes from the committed collection and is only quoted while the
    collection matches the run: same workload key, and kernel durations that agree (profile_check)."""
    other = None
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")), reverse=True):
        tj, ks, wk = os.path.join(d, "traffic.json"), os.path.join(d, "kernel_stats.txt"), os.path.join(d, "workload.json")
        if not (os.path.exists(tj) and os.path.exists(ks)):
            continue
        try:
            key = json.load(open(wk)) if os.path.exists(wk) else {"workload": "ce11cb4", "target_mbp": 100.0, "chunk": 250000}
            mine = {"workload": args.workload, "chunk": args.chunk,
                    "target_mbp": float(args.target_mbp or {"human": 500.0, "plumbing": 1.0}.get(args.workload, 100.0))}
            if any(key.get(k) != v for k, v in mine.items()) or args.target_fasta:
                other = other or os.path.relpath(tj, ROOT) + " (other workload: not quoted)"
                continue
            stats = {}
            for line in open(ks):
                m = re.match(r"\s+(?:sa::)?(\w+).*calls=(\d+) total_us=([\d.]+) avg_us=([\d.]+)", line)
                if m and m.group(1) not in stats:
                    stats[m.group(1)] = (int(m.group(2)), float(m.group(4)))
            stats["__single_stream__"] = bool(key.get("single_stream"))
            return json.load(open(tj)), os.path.relpath(tj, ROOT), stats
        except Exception:
            continue
    return None, other, None


def profile_check(prof, solo, kstats, args, scope_kernels=SCOPE_KERNELS):
    """Does the committed rocprofv3 collection describe the kernels of THIS run?  (With several calls in flight a kernel's
    duration depends on what it overlaps with, which a tracer perturbs: collections made with ONE call in flight -- workload.json
    "single_stream" -- are compared with this run's own single-stream launches instead.)  Per scope, the committed
    --kernel-trace --stats average (sum over the scope's kernels) is compared with the average this run's own HIP events
    predict for the same command: warmup + timed launches at the timed region's (concurrent) duration and the single-stream
    launches of the extra untimed pass.  The committed traffic is only quoted when they agree within 10 % (else the
    collection is stale: a kernel or the launch shape changed)."""
    if not kstats:
        return {"ok": False, "reason": "no committed kernel_stats for this workload"}
    out, ok = {}, True
    single = bool(kstats.get("__single_stream__"))  # the committed stats pass ran with one call in flight: compare like with like
    for scope in ("extend_filter", "extend_filter2", "seed_probe", "seed_lookup", "expand_hits"):
        if scope not in prof or not prof[scope][1]:
            continue
        committed = sum(kstats[k][1] for k in scope_kernels[scope] if k in kstats)
        scale = (args.steps + args.warmup) / max(args.steps, 1)
        s_ms, s_n = solo.get(scope, (0.0, 0))
        if single:
            if not s_n:
                continue
            ev = 1e3 * s_ms / s_n
        else:
            ev = 1e3 * (prof[scope][0] * scale + s_ms) / (prof[scope][1] * scale + s_n)
        out[scope] = {"events_us": round(ev, 2), "committed_us": round(committed, 2) if committed else None,
                      "ratio": round(ev / committed, 3) if committed else None}
        if not committed or abs(ev / committed - 1.0) > 0.10:
            ok = False
    return {"ok": ok and bool(out), "tolerance": 0.10, "scopes": out,
            "compared": "single-stream launches of this run vs a single-stream collection" if single else
                        "this run's launch mix (timed + warmup concurrent, extra pass single-stream) vs the same command under rocprofv3"}


# ------------------------------------------------------------------------------------------------------------------
def dry_run(args, rank, world, dist, torch, shard):
    """Everything of the bench contract that does not need a GPU: the REAL interval plan, rank walk and chunk arithmetic
    around a stub engine whose 'HSP count' is a checksum of the chunk bounds it was handed -- a wrong shard, walk or
    reduction changes the output."""
    qlen = int((args.target_mbp or 100.0) * 1e6)
    seed_size = len(SHAPE)
    items = shard.plan_intervals(qlen, seed_size, args.interval)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    bases, check = 0, 0
    for k in range(args.steps):
        for i, iv in enumerate(my_intervals(items, rank, k)):
            bases += iv[1] - iv[0]
            for rev in (False, True):  # the stub engine: "HSPs" = a checksum of the chunk bounds, weighted by the walk position
                for (a, b) in shard.chunks_of(iv, args.chunk, qlen - seed_size, rev):
                    check += (i + 1) * ((a * 31 + b * 17 + int(rev)) % 1000003)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([bases, check], dtype=torch.int64)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        bases, check = int(tb[0].item()), int(tb[1].item())
    if rank == 0:
        print(json.dumps({"metric": "Gbp query seeded+filtered+extended per sec", "value": bases / max(elapsed, 1e-9) / 1e9,
                          "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "int32", "data": "dry-run", "config": {"workload": "dry-run"},
                          "bases": bases, "checksum": check}))
    if dist is not None:
        dist.destroy_process_group()


def default_sub_mat(xdrop):
    """HOXD70 + L/N/X/E rows exactly as src/main.cpp:187-268 builds them for the default --ambiguous=x."""
    m = np.zeros((8, 8), dtype=np.int32)
    m[:4, :4] = [[91, -114, -31, -123], [-114, 100, -125, -31], [-31, -125, 100, -114], [-123, -31, -114, 91]]
    m[:4, 4] = m[4, :4] = -1000; m[4, 4] = -1000          # lower case (L)
    m[:5, 5] = m[5, :5] = -1000; m[5, 5] = -1000          # N
    m[:4, 6] = m[6, :4] = -100; m[4:6, 6] = m[6, 4:6] = -1000; m[6, 6] = -100  # X
    m[:, 7] = m[7, :] = -10 * xdrop                       # E ('&')
    return m.reshape(64)


def cpu_baseline(E, target, query, sub_mat, seed_size, kmer, args, xdrop, hspthresh, transition):
    """The oracle (a port: the reference cannot be compiled here and LASTZ is absent) timed on the host cores on a
    bounded sample of the SAME workload: whole 250 kbp chunks, both strands, until ~cpu_seconds have been spent.
    The seed table is copied from the device (it is parity-tested; building it on one CPU core takes longer than
    the whole budget) -- table build is outside the metric on both sides."""
    from oracle import oracle as O
    O.build(with_ref=False)
    cores = os.cpu_count() or 1
    O.generate_shape_pos(SHAPE)
    index = E.copy_index_table()
    pos = E.copy_pos_table()
    ref_codes = E.copy_ref_codes()
    q_codes = E.copy_query_codes(0, False)
    qrc_codes = E.copy_query_codes(0, True)
    qb = query.tobytes()
    qrc = O.rev_comp_ascii(qb, 0, query.size)
    done_bases, spent, chunks = 0, 0.0, 0
    end_pos = query.size - seed_size
    c = 0
    while spent < args.cpu_seconds and c < end_pos:
        e = min(c + args.chunk, end_pos)
        t0 = time.perf_counter()
        for rev, buf, codes in ((False, qb, q_codes), (True, qrc, qrc_codes)):
            a, b = (c, e) if not rev else (end_pos - e, end_pos - c)
            seeds = O.make_seeds(buf, 0, a, b, seed_size, kmer, transition)
            O.seed_and_filter(ref_codes, codes, index, pos, seeds, sub_mat, seed_size=seed_size, xdrop=xdrop,
                              hspthresh=hspthresh, noentropy=False, num_threads=cores)
        spent += time.perf_counter() - t0
        done_bases += e - c
        chunks += 1
        c = e
    out = {"value": round(done_bases / spent / 1e9, 6), "unit": "Gbp/s", "cores": cores, "kind": "port",
           "sample": "%d x %d bp query chunks, both strands, vs the full target (%.1f s CPU wall); host seeding loop + "
                     "OpenMP extension of oracle/segalign_oracle.c" % (chunks, args.chunk, spent)}
    lz = lastz_row(target, query, args)
    if lz:
        out["lastz"] = lz
    return out


def lastz_row(target, query, args):
    """when a `lastz` binary is on PATH, additionally time
    `lastz T[multiple] Q --seed=12of19 --hspthresh=3000 --xdrop=910 --nogapped` (single process) on a bounded sample --
    the first 1 Mbp of the target and of the query (the size of configs[0]).  LASTZ's HSP set differs from SegAlign's by
    design; this row is a time reference only.  None when there is no binary (the case on this image)."""
    exe = shutil.which("lastz")
    if not exe:
        return None
    d = tempfile.mkdtemp(prefix="sa_lastz_")
    try:
        n = 1_000_000
        paths = []
        for nm, seq in (("t", target), ("q", query)):
            p = os.path.join(d, nm + ".fa")
            with open(p, "wb") as f:
                for i, rec in enumerate(bytes(seq[:n]).split(b"&")):
                    f.write(b">%s%d\n" % (nm.encode(), i))
                    for j in range(0, len(rec), 60):
                        f.write(rec[j:j + 60] + b"\n")
            paths.append(p)
        t0 = time.perf_counter()
        r = subprocess.run([exe, paths[0] + "[multiple]", paths[1], "--seed=12of19", "--hspthresh=3000", "--xdrop=910", "--nogapped",
                            "--format=general:name1,start1,end1,name2,start2,end2,strand2,score"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        dt = time.perf_counter() - t0
        return {"value": round(min(n, query.size) / dt / 1e9, 6), "unit": "Gbp/s", "cores": 1, "kind": "lastz",
                "sample": "first %d bp of target x first %d bp of query, --nogapped (%.1f s, rc=%d, %d HSP lines)" %
                          (min(n, target.size), min(n, query.size), dt, r.returncode, r.stdout.count(b"\n"))}
    except Exception as e:  # a reference time only: never fail the bench over it
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
