"""Work decomposition of the reference host, restated as pure functions so that every rank derives the same shards.

  intervals : src/main.cpp:383-393   query block [0, len - seed_size) cut into lastz_interval pieces (10 Mbp)
  chunks    : src/seeder.cpp:48-51   each interval cut into wga_chunk pieces (250 kbp); the minus strand walks the
              src/seeder.cpp:33-34,89-91   same interval in reverse-complement coordinates
  shards    : SURVEY.md 8(e)         intervals are independent -> rank r of N takes intervals r, r+N, ...  No collective.
"""


def plan_intervals(block_len, seed_size, interval):
    end_pos = block_len - seed_size
    return [(s, min(s + interval, end_pos)) for s in range(0, max(end_pos, 0), interval)]


def chunks_of(iv, chunk, q_block_len, rev):
    """(start, end) chunk bounds of one interval on one strand, in that strand's coordinates."""
    s, e = iv
    if rev:
        s, e = q_block_len - e, q_block_len - s
    return [(c, min(c + chunk, e)) for c in range(s, e, chunk)]


def shard(items, rank, world):
    return list(items[rank::world])


def strand_chunks(intervals, q_block_len, chunk, rev):
    """Every wga_chunk piece of one strand of a query block, ascending in that strand's coordinates: (interval index, a, b).  The
    minus strand walks the intervals in reverse (src/seeder.cpp:33-34,89-91)."""
    out = []
    order = range(len(intervals) - 1, -1, -1) if rev else range(len(intervals))
    for idx in order:
        s, e = intervals[idx]
        a, b = (q_block_len - e, q_block_len - s) if rev else (s, e)
        out.extend((idx, c, min(c + chunk, b)) for c in range(a, b, chunk))
    return out


def call_jobs(intervals, q_block_len, chunk, chunks_per_call=16):
    """The SeedAndFilter CALLS of one pass over a query block: per strand the wga_chunk pieces of all intervals (src/seeder.cpp:48-51,
    :89-91 in reverse-complement coordinates) are grouped into calls of at most `chunks_per_call` consecutive chunks -- equal groups
    (40 chunks with a limit of 16 go as 14 + 14 + 12).  A call is a range [a, b) that the engine cuts into chunks from `a` on, so a
    group only continues over pieces that touch and are full-sized except the last: groups run across interval borders where the
    chunk grid does (10 Mbp intervals hold 40 chunks exactly) and end at a short last chunk (the tail interval, which comes FIRST on
    the minus strand).  One dict per call: first interval, strand, [a, b) in that strand's coordinates, chunks.  Every call is
    independent (own iteration plans, own dedup scopes; a chunk's output belongs to the interval it lies in, whatever call carried
    it), so ANY assignment of calls to GPUs gives identical bytes."""
    jobs = []
    for rev in (False, True):
        pieces = strand_chunks(intervals, q_block_len, chunk, rev)
        runs, cur = [], []   # maximal runs a single range can describe
        for p in pieces:
            if cur and (cur[-1][2] != p[1] or cur[-1][2] - cur[-1][1] != chunk):
                runs.append(cur)
                cur = []
            cur.append(p)
        if cur:
            runs.append(cur)
        for run in runs:
            n = len(run)
            ncalls = (n + chunks_per_call - 1) // chunks_per_call
            group = (n + ncalls - 1) // ncalls
            for g in range(0, n, group):
                part = run[g:g + group]
                jobs.append(dict(interval=part[0][0], rev=rev, a=part[0][1], b=part[-1][2], chunks=len(part)))
    return jobs


def partition(jobs, rank, world, weights=None, offset=0):
    """Strong scaling: the calls of ONE problem dealt to the ranks -- every call exactly once, no communication: every rank computes
    the same map.  Without weights: round-robin (consecutive calls -- neighbouring query regions, similar hit density -- land on
    different ranks; `offset` continues the deal from one pass to the next).  With weights (the seed hits of every call, counted by an untimed pass that every rank runs identically):
    longest-processing-time-first -- calls in descending weight, each to the rank with the least weight so far (ties: lowest rank)
    -- which bounds the imbalance by one call's weight.  A rank's calls keep their original order."""
    if weights is None or world <= 1:
        # offset: how many calls have been dealt before this list -- the deal of pass k continues where pass k - 1 stopped (a host
        # that walks query block after query block keeps dealing; 20 calls on 8 ranks are 3/3/3/3/2/2/2/2 in one pass and even
        # over two), like the reference's dynamic device pool (src/seed_filter.cu:699-706)
        return [j for k, j in enumerate(jobs) if (k + offset) % world == rank]
    assert len(weights) == len(jobs)
    load = [0] * world
    owner = [0] * len(jobs)
    for k in sorted(range(len(jobs)), key=lambda i: (-int(weights[i]), i)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[k] = r
        load[r] += int(weights[k])
    return [j for k, j in enumerate(jobs) if owner[k] == rank]


def hsp_checksum(segs, rev):
    """Order-independent checksum of a list of HSP records (numpy structured array ref_start, query_start, len, score): the sum
    over records of a 61-bit mix of the fields and the strand.  Equal HSP multisets <=> equal sums (up to hash collisions), so
    the N-GPU run of a problem must reproduce the 1-GPU checksum whatever the partition."""
    import numpy as np
    if segs is None or len(segs) == 0:
        return 0
    x = (segs["ref_start"].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (segs["query_start"].astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
    x ^= (segs["len"].astype(np.uint64) << np.uint64(17)) ^ (segs["score"].astype(np.int64).astype(np.uint64) * np.uint64(0x165667B19E3779F9))
    x ^= np.uint64(0xD6E8FEB86659FD93) if rev else np.uint64(0)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(32)
    return int(np.sum(x & np.uint64((1 << 61) - 1), dtype=np.uint64) & np.uint64((1 << 63) - 1))


def rm_plan(seq_len, seq_block_size=1000000000, lastz_interval_size=10000000, prop_neigh_interval=0.2, seed_size=19):
    """repeat_masker_src/main.cpp:316-436 (float/ceil arithmetic included): one dict per (block, interval) task with the
    seed range [start, end) and the target window [ref_start, ref_end] inside the block."""
    import math

    import numpy as np
    f32 = np.float32
    if seq_block_size == 1000000000:
        seq_block_size -= seq_block_size % lastz_interval_size            # :255-258
    total_query_intervals = int(math.ceil(f32(seq_len) / f32(lastz_interval_size)))  # :316
    num_neigh_interval = int(math.ceil(f32(prop_neigh_interval) * f32(total_query_intervals)))
    left_intervals = int(math.ceil(f32((num_neigh_interval - 1) & 0xFFFFFFFF) / f32(2)))  # :319 (unsigned arithmetic)
    right_intervals = (num_neigh_interval - 1 - left_intervals) & 0xFFFFFFFF
    left_overlap = (left_intervals * lastz_interval_size) & 0xFFFFFFFF
    right_overlap = (right_intervals * lastz_interval_size) & 0xFFFFFFFF
    max_interval_seq_len = (left_overlap + lastz_interval_size + right_overlap) & 0xFFFFFFFF
    tasks, block_index, l = [], 0, 0
    while l < seq_len:                                                     # :341
        block_start = l if l < left_overlap else l - left_overlap
        if l + seq_block_size + right_overlap > seq_len:
            block_len = seq_len - block_start
        else:
            block_len = (((l - block_start + seq_block_size) & 0xFFFFFFFF) + right_overlap) & 0xFFFFFFFF  # (:358: unsigned int arithmetic)
        start_pos = l - block_start
        if block_len < seq_block_size:
            end_pos = start_pos + block_len - (l - block_start) - seed_size
        else:
            end_pos = start_pos + seq_block_size - seed_size
        while start_pos < end_pos:                                         # :367
            end = min(end_pos, start_pos + lastz_interval_size)
            left_limit = start_pos < left_overlap
            right_limit = ((end + right_overlap) & 0xFFFFFFFF) > block_len       # (:385: uint32 -- wraps with neighbor_proportion 0)
            if left_limit:
                ref_start = 0
                ref_end = block_len if right_limit else min(max_interval_seq_len, block_len)
            elif right_limit:
                ref_end = block_len
                ref_start = 0 if block_len < max_interval_seq_len else block_len - max_interval_seq_len
            else:
                ref_start, ref_end = (start_pos - left_overlap) & 0xFFFFFFFF, (end + right_overlap) & 0xFFFFFFFF
            tasks.append(dict(block_index=block_index, block_start=block_start, block_len=block_len, start=start_pos, end=end,
                              ref_start=ref_start, ref_end=ref_end))
            start_pos += lastz_interval_size
        block_index += 1
        l += seq_block_size
    return tasks
