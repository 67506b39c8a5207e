"""FASTA -> sequence blocks the way the reference host lays them out (src/main.cpp:320-409,475-549): the records of a
file are appended to one arena, separated by a single '&' inside a block; a block is closed as soon as it has grown beyond
SEQ_BLOCK_SIZE = 500 Mbp (src/graph.h:10, the test at src/main.cpp:359,515 runs after a record was added).  Used by
bench.py for --target-fasta / --query-fasta; the C++ harness (segalign_amd/host) has its own reader."""
import gzip

import numpy as np

SEQ_BLOCK_SIZE = 500_000_000  # src/graph.h:10


def read_records(path):
    """[(name, uint8 array)] of a plain or gzip FASTA file."""
    op = gzip.open if path.endswith(".gz") else open
    recs, name, parts = [], None, []
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, np.frombuffer(b"".join(parts), dtype=np.uint8)))
                name, parts = line[1:].split()[0].decode() if line[1:].split() else "", []
            elif name is not None:
                parts.append(line.rstrip(b"\r\n"))
    if name is not None:
        recs.append((name, np.frombuffer(b"".join(parts), dtype=np.uint8)))
    return recs


def plan_blocks(lengths, block_size=SEQ_BLOCK_SIZE):
    """Record index ranges [(first, last+1)] of the blocks: records accumulate until the block exceeds block_size."""
    blocks, first, acc = [], 0, 0
    for i, n in enumerate(lengths):
        acc += n + (1 if i > first else 0)  # the '&' in front of every record but the first of a block
        if acc > block_size:
            blocks.append((first, i + 1))
            first, acc = i + 1, 0
    if first < len(lengths):
        blocks.append((first, len(lengths)))
    return blocks


def block_bytes(recs, rng):
    amp = np.frombuffer(b"&", dtype=np.uint8)
    parts = []
    for k in range(rng[0], rng[1]):
        if k > rng[0]:
            parts.append(amp)
        parts.append(recs[k][1])
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)


def first_block(path, block_size=SEQ_BLOCK_SIZE):
    recs = read_records(path)
    blocks = plan_blocks([r[1].size for r in recs], block_size)
    return block_bytes(recs, blocks[0]), len(blocks)
