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
