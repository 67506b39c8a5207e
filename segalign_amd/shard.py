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
            block_len = l - block_start + seq_block_size + right_overlap
        start_pos = l - block_start
        if block_len < seq_block_size:
            end_pos = start_pos + block_len - (l - block_start) - seed_size
        else:
            end_pos = start_pos + seq_block_size - seed_size
        while start_pos < end_pos:                                         # :367
            end = min(end_pos, start_pos + lastz_interval_size)
            left_limit = start_pos < left_overlap
            right_limit = end + right_overlap > block_len
            if left_limit:
                ref_start = 0
                ref_end = block_len if right_limit else min(max_interval_seq_len, block_len)
            elif right_limit:
                ref_end = block_len
                ref_start = 0 if block_len < max_interval_seq_len else block_len - max_interval_seq_len
            else:
                ref_start, ref_end = start_pos - left_overlap, end + right_overlap
            tasks.append(dict(block_index=block_index, block_start=block_start, block_len=block_len, start=start_pos, end=end,
                              ref_start=ref_start, ref_end=ref_end))
            start_pos += lastz_interval_size
        block_index += 1
        l += seq_block_size
    return tasks
