#!/usr/bin/env python3
"""BASELINE configs[2] walked as ONE problem on one GPU: a G x G grid of 500 Mbp target blocks x 500 Mbp query blocks the way
src/main.cpp:601-737 walks it -- per target block: g_ClearRef, upload, GenerateSeedPosTable (a 165-230 GB context table rebuilt
in the warm arena); per query block: upload into the OTHER device buffer (BUFFER_DEPTH = 2) by a background thread while the
calls of the current block run; the calls of a block pair through sa_seed_calls (six in flight).

    python tools/human_grid.py --grid 2            # 2 x 2 blocks of 500 Mbp
    python tools/human_grid.py --grid 6            # the 3 Gbp x 3 Gbp problem

Target block i = synth.human_target_block(i); query block j = pieces of target block j, 1.2 % diverged (synth.human_query_block):
the diagonal of the grid holds the homologous pairs, the rest only chance hits.  Block pair (0, 0) is exactly what
`bench.py --workload human --query-mbp <same>` runs, so its HSP checksum must equal that line's (--check-bench runs it).
Prints one JSON document: per target block upload + table build wall time (block 0 cold, the others in the warm arena), per block
pair seconds / HSPs / checksum / seed hits, the grid's whole wall time and query-bases x target-blocks per second."""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SHAPE = "TTT0T00TT00T0T0TTTT"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=2)
    ap.add_argument("--target-mbp", type=float, default=500.0)
    ap.add_argument("--query-mbp", type=float, default=500.0)
    ap.add_argument("--in-flight", type=int, default=6)
    ap.add_argument("--check-bench", action="store_true", help="run bench.py --workload human on block pair (0, 0) and compare checksums")
    ap.add_argument("--max-hits-mem-gb", type=float, default=None,
                    help="walk the grid under the MAX_HITS of a reference GPU of this many GiB (src/seed_filter.cu:832-841; 8 = the M60 of README.md:27)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench  # default_sub_mat, CHECK_MOD
    from segalign_amd import engine as E
    from segalign_amd import shard, synth

    G = args.grid
    tlen, qlen = int(args.target_mbp * 1e6), int(args.query_mbp * 1e6)
    os.environ.setdefault("SEGALIGN_AMD_SLOTS", str(args.in_flight))
    os.environ.setdefault("SEGALIGN_AMD_ARENA_GB", "180")
    E.select_devices([0])
    E.InitializeInterface(1)
    kmer = E.GenerateShapePos(SHAPE)
    E.InitializeProcessor(True, 250000, 19, bench.default_sub_mat(910), 910, 3000, False)
    if args.max_hits_mem_gb:
        E.set_max_hits(E.max_hits_for_mem(int(args.max_hits_mem_gb * (1 << 30))))

    t_gen0 = time.time()
    targets = [None] * G
    queries = []
    for j in range(G):   # query block j goes with target block j; targets are regenerated when their turn comes (500 MB each)
        t = synth.human_target_block(tlen, j)
        queries.append(synth.human_query_block(t, qlen, j))
        if j == 0:
            targets[0] = t
        del t
    t_gen = time.time() - t_gen0

    blocks, pairs = [], []
    next_target = {}

    def gen_target(i):
        next_target[i] = synth.human_target_block(tlen, i)

    grid_t0 = time.time()
    compute_s = 0.0
    for i in range(G):
        target = targets[0] if i == 0 else next_target.pop(i)
        prefetch = None
        if i + 1 < G:   # the host produces the next target block while this one is worked on (the reference reads FASTA meanwhile)
            prefetch = threading.Thread(target=gen_target, args=(i + 1,))
            prefetch.start()
        b = dict(target_block=i, bases=int(target.size))
        t0 = time.time()
        if i > 0:
            E.ClearRef()
        b["clear_ref_s"] = round(time.time() - t0, 4)
        t0 = time.time()
        keep = E.SendRefWriteRequest(target, 0, target.size)
        b["upload_encode_s"] = round(time.time() - t0, 4)
        t0 = time.time()
        E.GenerateSeedPosTable(keep, 0, target.size, 1, 19, kmer)
        b["table_build_s"] = round(time.time() - t0, 4)
        b["lookup_mode"] = int(E.lib().sa_get_lookup_mode())
        b["neighbourhood_entries"] = int(E.lib().sa_get_neighbourhood_entries())
        b["chunks_per_call"] = int(E.lib().sa_get_chunks_per_call())
        blocks.append(b)
        # query blocks: block 0 of this target goes up now, block j + 1 while block j runs
        t0 = time.time()
        E.SendQueryWriteRequest(queries[0], 0, queries[0].size, 0)
        first_upload = time.time() - t0
        for j in range(G):
            buf = j % 2
            up = None
            up_s = [0.0]
            if j + 1 < G:
                def upload(nj=j + 1):
                    tu = time.time()
                    E.SendQueryWriteRequest(queries[nj], 0, queries[nj].size, nj % 2)
                    up_s[0] = time.time() - tu
                up = threading.Thread(target=upload)
                up.start()
            q = queries[j]
            q_block_len = q.size - 19
            intervals = shard.plan_intervals(q.size, 19, 10_000_000)
            jobs = shard.call_jobs(intervals, q_block_len, 250000, b["chunks_per_call"])
            hits = []
            t0 = time.time()
            outs, st = E.SeedCalls([(c["a"], c["b"], c["rev"]) for c in jobs], buf, args.in_flight, hits_out=hits)
            dt = time.time() - t0
            compute_s += dt
            chk = 0
            for c, o in zip(jobs, outs):
                chk = (chk + shard.hsp_checksum(o, c["rev"])) % bench.CHECK_MOD
            if up is not None:
                up.join()
            pairs.append(dict(target_block=i, query_block=j, seconds=round(dt, 4), gbp_per_s=round(q_block_len / dt / 1e9, 4), calls=len(jobs),
                              hsps=int(sum(o.size for o in outs)), seed_hits=int(sum(hits)), hsp_checksum=chk,
                              next_query_upload_s=round(up_s[0], 4), first_query_upload_s=round(first_upload, 4) if j == 0 else None))
            sys.stderr.write("pair (%d, %d): %.3f s, %d HSPs\n" % (i, j, dt, pairs[-1]["hsps"]))
        if prefetch is not None:
            prefetch.join()
        del keep, target
    grid_s = time.time() - grid_t0
    q_bases = sum(int(q.size) - 19 for q in queries)
    warm = [b["table_build_s"] for b in blocks[1:]]
    doc = dict(workload="BASELINE configs[2] stand-in: %d x %d grid of %.0f Mbp target blocks x %.0f Mbp query blocks, 12of19 + transitions, one MI355X"
                        % (G, G, tlen / 1e6, qlen / 1e6),
               max_hits=int(E.get_max_hits()),
               generate_s=round(t_gen, 2), grid_wall_s=round(grid_s, 3), compute_s=round(compute_s, 3),
               non_scaling_s=round(sum(b["clear_ref_s"] + b["upload_encode_s"] + b["table_build_s"] for b in blocks), 3),
               query_bases_x_target_blocks=q_bases * G, gbp_per_s=round(q_bases * G / grid_s / 1e9, 4),
               table_build_cold_s=blocks[0]["table_build_s"], table_build_warm_s=warm,
               blocks=blocks, pairs=pairs)
    # what an N-GPU walk of this grid would take by these numbers: every rank rebuilds every target block (no collective), the calls
    # of a block pair are dealt to the ranks
    per_block_fixed = [b["clear_ref_s"] + b["upload_encode_s"] + b["table_build_s"] for b in blocks]
    doc["projection"] = {str(n): round(grid_s / (sum(per_block_fixed) + compute_s / n), 2) for n in (1, 2, 4, 8)}
    doc["projection_note"] = "speed-up over this run at N ranks = wall / (fixed per-target-block time + compute / N): the replicated table build is the non-scaling term"
    if args.check_bench:
        E.ShutdownProcessor()
        E.lib().sa_release_arena()
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "human", "--query-mbp", str(args.query_mbp), "--target-mbp",
                              str(args.target_mbp), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-dropin", "--no-roofline"] +
                             (["--max-hits-mem-gb", str(args.max_hits_mem_gb)] if args.max_hits_mem_gb else []),
                             cwd=ROOT, capture_output=True, text=True, timeout=1800)
        line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
        if out.returncode != 0 or not line:
            doc["bench_check"] = dict(ok=False, error=out.stderr[-1500:])
        else:
            d = json.loads(line[-1])
            doc["bench_check"] = dict(ok=d["config"]["hsp_checksum"] == pairs[0]["hsp_checksum"] and d["config"]["hsps_per_step"] == pairs[0]["hsps"],
                                      bench_checksum=d["config"]["hsp_checksum"], grid_checksum=pairs[0]["hsp_checksum"],
                                      bench_hsps=d["config"]["hsps_per_step"], grid_hsps=pairs[0]["hsps"], bench_value=d["value"], bench_ms_per_step=d["ms_per_step"])
    else:
        E.ShutdownProcessor()
    text = json.dumps(doc, indent=1)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
