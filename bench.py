#!/usr/bin/env python3
"""bench.py -- Gbp of query seeded + filtered + extended per second on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  One JSON line on stdout from rank 0.

Workload = BASELINE.json configs[1] (ce11 x cb4, default 12of19 seed, 1 x MI355X).  The real assemblies cannot
be downloaded here, so the stand-in of BASELINE.md / SURVEY.md 8(d) is generated deterministically: a
~100 Mbp, 7-record target and a query that is an 8 %-diverged copy with 20 % soft-masked runs and inversions.
A *step* is one 10 Mbp query interval (the reference's lastz_interval, src/graph.h:11) on BOTH strands against
the resident target: 40 + 40 SeedAndFilter calls of 250 kbp (DEFAULT_WGA_CHUNK), seeds generated on the device
 so the whole "seeded + filtered + extended" metric is inside the timed region and no seed vector
crosses PCIe.  Target upload, encoding and the seed-table build happen before the timed region (reported
separately, as the reference does under --debug, src/main.cpp:617-629).

Multi-GPU: query intervals are independent shards: rank r takes intervals r, r+N, ...; every rank
holds the target + table; there is NO data-path collective.  Per-GPU work is fixed => "scaling": "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = "TTT0T00TT00T0T0TTTT"  # 12of19, src/main.cpp:160-163
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--target-mbp", type=float, default=100.0, help="synthetic target size (ce11 ~ 100.3 Mbp)")
    ap.add_argument("--interval", type=int, default=10_000_000)
    ap.add_argument("--chunk", type=int, default=250_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU baseline budget")
    ap.add_argument("--host-threads", type=int, default=2,
                    help="host threads issuing SeedAndFilter calls (the reference runs one TBB seeder body per core; "
                         "the engine has 2 slots per device so one call's syncs overlap another call's kernels)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only rehearsal of the launch/shard/reduce/JSON contract (gloo, no engine, no GPU)")
    return ap.parse_args()


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
    from segalign_amd import synth

    # ---------------- workload: ce11 x cb4 stand-in ----------------
    t_gen0 = time.time()
    tlen = int(args.target_mbp * 1e6)
    target, query = synth.make_pair(tlen, 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, indel_every=0,
                                    invert_frac=0.3, invert_block=100_000)
    t_gen = time.time() - t_gen0

    # default parameters of the reference (src/main.cpp:61-124)
    xdrop, hspthresh, seed_size = 910, 3000, len(SHAPE)
    sub_mat = default_sub_mat(xdrop)

    E.select_devices([local_rank])
    E.InitializeInterface(1)
    kmer = E.GenerateShapePos(SHAPE)
    E.InitializeProcessor(True, args.chunk, seed_size, sub_mat, xdrop, hspthresh, False)
    t0 = time.time()
    keep = E.SendRefWriteRequest(target, 0, target.size)
    t_ref = time.time() - t0
    t0 = time.time()
    E.GenerateSeedPosTable(keep, 0, target.size, 1, seed_size, kmer)
    t_table = time.time() - t0
    t0 = time.time()
    E.SendQueryWriteRequest(query, 0, query.size, 0)
    t_query = time.time() - t0

    # intervals like src/main.cpp:383-393 over [0, len - seed_size)
    intervals = shard.plan_intervals(query.size, seed_size, args.interval)
    q_block_len = query.size - seed_size  # q_len handed to the seeder (main.cpp:708)

    def run_interval(iv, collect=None):
        """one step: seeder_body::operator() for the interval (src/seeder.cpp:12-127) -- every 250 kbp chunk of both
        strands through the engine, `host_threads` chunk calls in flight, issued by the library's own C++ threads"""
        fw, rc, st = E.SeedInterval(iv[0], iv[1], q_block_len, E.STRAND_BOTH, 0, max(1, args.host_threads))
        if collect is not None:
            collect.append(st)
        return iv[1] - iv[0], int(fw.size + rc.size)

    my = shard.shard(intervals, rank, world) or intervals

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- warmup (untimed) ----------------
    for w in range(args.warmup):
        run_interval(my[w % len(my)])

    # ---------------- timed region ----------------
    E.profile_reset()
    E.profile_enable(True)
    call_stats = []
    barrier()
    t0 = time.perf_counter()
    bases = 0
    hsps = 0
    for k in range(args.steps):
        b, h = run_interval(my[k % len(my)], call_stats)
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

    # ---------------- roofline of the dominant kernel (rank 0's own launches) ----------------
    roof = None
    if rank == 0 and prof:
        # the same kernel without a second call overlapping it: one extra (untimed) interval issued by ONE host thread
        saved_threads, args.host_threads = args.host_threads, 1
        E.profile_reset()
        E.profile_enable(True)
        solo_stats = []
        run_interval(my[0], solo_stats)
        E.profile_enable(False)
        solo = E.profile_entries()
        args.host_threads = saved_threads
        # per-hit ratios from one instrumented (untimed) interval: deterministic, identical work
        E.set_count_examined(True)
        sample = []
        run_interval(my[0], sample)
        E.set_count_examined(False)
        sH = max(sum(s["num_hits"] for s in sample), 1)
        e_all = sum(s["num_examined"] for s in sample) / sH          # E per hit (reference algorithm)
        e_flt = sum(s["num_examined_filter"] for s in sample) / sH   # bases per hit scored by the filter kernel
        H = sum(s["num_hits"] for s in call_stats)
        A = sum(s["num_survivors"] for s in call_stats)
        S = sum(s["num_seeds"] for s in call_stats)
        Cn = sum(s["num_candidates"] for s in call_stats)
        kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1], "avg_us": round(1e3 * v[0] / max(v[1], 1), 2)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}

        def gbs(nbytes, keys):
            ms = sum(prof[k][0] for k in keys if k in prof)
            return (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else None

        # algorithmic bytes (restated per kernel in DESIGN.md):
        #   extension as a whole (filter+exact+entropy) : 8H + 2E + 20A
        #   X-drop filter alone                         : 8H + 2*E_filter + 12*C   (C candidates out)
        #   lookup + expand                             : 16S + 12H
        ext_bytes = 8.0 * H + 2.0 * e_all * H + 20.0 * A
        flt_bytes = 8.0 * H + 2.0 * e_flt * H + 12.0 * Cn
        look_bytes = 16.0 * S + 12.0 * H
        name, (ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
        per_kernel = {"extend_filter": flt_bytes, "seed_lookup": 16.0 * S, "expand_hits": 12.0 * H}
        achieved = gbs(per_kernel[name], [name]) if name in per_kernel else None
        ext_gbs = gbs(ext_bytes, ["extend_filter", "extend_exact", "extend_entropy"])
        look_gbs = gbs(look_bytes, ["seed_lookup", "expand_hits"])
        traffic, traffic_src = measured_traffic(name, E.filter_mode())
        single = None
        if name == "extend_filter" and name in solo and solo[name][1]:
            sH = sum(x["num_hits"] for x in solo_stats)
            sC = sum(x["num_candidates"] for x in solo_stats)
            sb = 8.0 * sH + 2.0 * e_flt * sH + 12.0 * sC
            sg = sb / (solo[name][0] * 1e-3) / 1e9
            single = {"avg_launch_us": round(1e3 * solo[name][0] / solo[name][1], 2), "achieved": round(sg, 1),
                      "frac": round(sg / HBM_PEAK_GBS, 4),
                      "note": "same kernel, one call in flight (no overlap with a second stream); untimed extra interval"}
        roof = {
            "bound": "hbm", "kernel": name, "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
            "traffic_source": traffic_src,
            # measured fabric traffic over the single-stream launch time: how close the kernel runs to the HBM limit in
            # bytes it really moves (every 16-byte random window costs a 128-byte line, which `achieved` does not count)
            "traffic_gbs_single_stream": (round(traffic / (single["avg_launch_us"] * 1e-6) / 1e9, 1)
                                          if (traffic and single) else None),
            # the same traffic counted in 128-byte lines against the measured random-gather ceiling of the chip
            # (tools/micro/gather_bw.hip: 57 G lines/s for a 100 MB working set) -- the bound that actually applies
            "random_line_roofline": ({"lines_per_launch": int(traffic // 128), "peak_lines_per_s": RANDOM_LINES_PER_S,
                                      "frac_single_stream": round(traffic / 128 / (single["avg_launch_us"] * 1e-6) / RANDOM_LINES_PER_S, 4)}
                                     if (traffic and single) else None),
            "kernel_symbol": FILTER_KERNELS.get(E.filter_mode()) if name == "extend_filter" else None,
            "calls_in_flight": max(1, args.host_threads), "single_stream": single,
            "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "algorithmic_bytes_per_launch": round(per_kernel.get(name, 0) / max(launches, 1)),
            "per_hit": {"examined_bases_E": round(e_all, 2), "examined_by_filter": round(e_flt, 2),
                        "candidate_frac": round(Cn / max(H, 1), 5), "survivor_frac": round(A / max(H, 1), 5)},
            "extension_total": {"achieved": round(ext_gbs, 1) if ext_gbs else None,
                                "frac": round(ext_gbs / HBM_PEAK_GBS, 4) if ext_gbs else None, "bytes": "8*H + 2*E + 20*A"},
            "lookup_expand": {"achieved": round(look_gbs, 1) if look_gbs else None,
                              "frac": round(look_gbs / HBM_PEAK_GBS, 4) if look_gbs else None, "bytes": "16*S + 12*H",
                              # the same two kernels by the HBM traffic rocprofv3 measured (profiles/), over their
                              # single-stream launch time: every 8-byte bucket gather moves a 128-byte line, so the
                              # kernels sit near the roofline in bytes MOVED while `achieved` counts bytes NEEDED
                              "measured_traffic": measured_pair(solo)},
            "kernels": kernels,
        }

    # ---------------- CPU baseline (rank 0, N == 1 only, bounded sample) ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(E, target, query, sub_mat, seed_size, kmer, args, xdrop, hspthresh)

    if rank == 0:
        value = bases / elapsed / 1e9
        line = {
            "metric": "Gbp query seeded+filtered+extended per sec", "value": round(value, 5), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "ce11 x cb4 stand-in (BASELINE configs[1]): %.0f Mbp 7-record target x 8%%-diverged "
                                   "soft-masked query, 12of19 + transitions, HOXD70, xdrop 910, hspthresh 3000; "
                                   "step = one %d bp query interval, both strands, %d bp chunks (4 chunks of a strand share one "
                                   "pass over the kernels), device-side seeding" % (args.target_mbp, args.interval, args.chunk),
                       "parallelism": "query-interval shards x%d, no collective" % world,
                       "hsps_per_step": hsps // max(args.steps * world, 1)},
            "setup_s": {"generate": round(t_gen, 2), "target_upload_encode": round(t_ref, 3),
                        "seed_table_build": round(t_table, 3), "query_upload_encode": round(t_query, 3)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    E.ShutdownProcessor()
    if dist is not None:
        dist.destroy_process_group()


RANDOM_LINES_PER_S = 57e9  # measured on MI355X: random 128-byte line gathers per second (tools/micro/gather_bw.hip)
FILTER_KERNELS = {0: "extend_filter_kernel", 1: "extend_filter_kernel", 3: "extend_filter_packed_kernel"}  # sa_get_filter_mode() -> kernel behind the "extend_filter" scope


def measured_traffic(prof_name, filter_mode=3):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 correction: see tools/traffic_json.py).  bench.py cannot profile itself, so it quotes the
    newest profiles/rNN/traffic.json collected with tools/profile_bench.sh on this same default workload; null if absent."""
    import glob
    kernel = {"extend_filter": FILTER_KERNELS.get(filter_mode), "extend_exact": "extend_exact_kernel",
              "expand_hits": "expand_hits_kernel", "seed_lookup": "seed_lookup_kernel"}.get(prof_name)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not kernel or not files:
        return None, None
    try:
        t = json.load(open(files[-1]))
        return t[kernel]["hbm_bytes"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def measured_pair(solo):
    out = {}
    for name in ("seed_lookup", "expand_hits"):
        t, _ = measured_traffic(name)
        if t and name in solo and solo[name][1]:
            us = 1e3 * solo[name][0] / solo[name][1]
            out[name] = {"hbm_bytes_per_launch": int(t), "single_stream_us": round(us, 2),
                         "gbs": round(t / (us * 1e-6) / 1e9, 1), "frac": round(t / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                         "line_frac": round(t / 128 / (us * 1e-6) / RANDOM_LINES_PER_S, 4)}
    return out or None


def dry_run(args, rank, world, dist, torch, shard):
    """Everything of the bench contract that does not need a GPU: sharding, barrier, max/sum reduction, the JSON line.
    The per-interval 'work' is a checksum of the chunk bounds, so a wrong shard or reduction changes the output."""
    qlen = int(args.target_mbp * 1e6)
    seed_size = len(SHAPE)
    intervals = shard.plan_intervals(qlen, seed_size, args.interval)
    my = shard.shard(intervals, rank, world) or intervals
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    bases, check = 0, 0
    for k in range(args.steps):
        iv = my[k % len(my)]
        bases += iv[1] - iv[0]
        for rev in (False, True):
            for (a, b) in shard.chunks_of(iv, args.chunk, qlen - seed_size, rev):
                check += (a * 31 + b * 17 + int(rev)) % 1000003
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


def cpu_baseline(E, target, query, sub_mat, seed_size, kmer, args, xdrop, hspthresh):
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
            seeds = O.make_seeds(buf, 0, a, b, seed_size, kmer, True)
            O.seed_and_filter(ref_codes, codes, index, pos, seeds, sub_mat, seed_size=seed_size, xdrop=xdrop,
                              hspthresh=hspthresh, noentropy=False, num_threads=cores)
        spent += time.perf_counter() - t0
        done_bases += e - c
        chunks += 1
        c = e
    return {"value": round(done_bases / spent / 1e9, 6), "unit": "Gbp/s", "cores": cores, "kind": "port",
            "sample": "%d x %d bp query chunks, both strands, vs the full target (%.1f s CPU wall); host seeding loop + "
                      "OpenMP extension of oracle/segalign_oracle.c" % (chunks, args.chunk, spent)}


if __name__ == "__main__":
    main()
