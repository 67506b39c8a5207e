"""FASTA -> sequence blocks the way the reference host lays them out (src/main.cpp:320-409,475-549): the records of a
file are appended to one arena, separated by a single '&' inside a block; a block is closed as soon as it has grown beyond
SEQ_BLOCK_SIZE = 500 Mbp (src/graph.h:10, the test at src/main.cpp:359,515 runs after a record was added).  Used by
bench.py for --target-fasta / --query-fasta; the C++ harness (segalign_amd/host) has its own reader."""
import gzip

import numpy as np

SEQ_BLOCK_SIZE = 500_000_000  # src/graph.h:10


_SPACE = frozenset(b" \t\n\v\f\r")   # isspace()


def kseq_records(buf):
    """The records klib's kseq_read returns for the bytes of a file, read as the reference reads them (common/kseq.h:177-218, instantiated at
    src/main.cpp:21, looped over at :336 / :494 until kseq_read < 0) -> [(name bytes, sequence bytes)].  A FASTA / FASTQ state machine, not a
    line filter: the first '>' or '@' ANYWHERE opens the first header; the name ends at the first isspace(); a sequence runs until a LINE that
    starts with '>', '@' or '+'; empty lines are skipped; one trailing CR is taken off the accumulated sequence after every line (when it is
    longer than one character, :141); '+' opens a quality block that swallows lines until it is as long as the sequence, and a block of another
    length ends the whole read.  Pinned to the real header by tests/golden/kseq_golden.json (the header itself compiled as it lies: tests/test_fasta_kseq.py)."""
    n, p, last, out = len(buf), 0, 0, []
    while True:
        if last == 0:   # jump to the next header character (:182-186)
            i1, i2 = buf.find(b">", p), buf.find(b"@", p)
            c = min([i for i in (i1, i2) if i >= 0], default=-1)
            if c < 0:
                break
            p = c + 1
        if p >= n:      # ks_getuntil finds nothing and the stream is at its end (:188)
            break
        q = p
        while q < n and buf[q] not in _SPACE:
            q += 1
        name, c = buf[p:q], (buf[q] if q < n else 0)
        p = min(q + 1, n)
        if c != 10:     # the comment: the rest of the header line (:189)
            e = buf.find(b"\n", p)
            p = e + 1 if e >= 0 else n
        seq = bytearray()
        while True:     # :194-198
            if p >= n:
                c = -1
                break
            c = buf[p]
            p += 1
            if c in (62, 43, 64):
                break
            if c == 10:
                continue
            seq.append(c)
            if p >= n:      # ks_getuntil2 at the end of the stream returns before it looks for the CR (:139)
                continue
            e = buf.find(b"\n", p)
            seq += buf[p:e] if e >= 0 else buf[p:]
            p = e + 1 if e >= 0 else n
            if len(seq) > 1 and seq[-1] == 13:
                del seq[-1]
        if c in (62, 64):
            last = c
        if c != 43:
            out.append((bytes(name), bytes(seq)))
            continue
        e = buf.find(b"\n", p)   # FASTQ: skip the rest of the '+' line (:211-212)
        if e < 0:
            break
        p = e + 1
        qual = 0
        while True:     # :213: lines are swallowed until the quality is as long as the sequence
            if p >= n:
                break
            e = buf.find(b"\n", p)
            line = buf[p:e] if e >= 0 else buf[p:]
            p = e + 1 if e >= 0 else n
            qual += len(line)
            if qual > 1 and line[-1:] == b"\r":
                qual -= 1
            if qual >= len(seq):
                break
        last = 0
        if qual != len(seq):
            break
        out.append((bytes(name), bytes(seq)))
    return out


def read_records(path):
    """[(name, uint8 array)] of a plain or gzip FASTA file, as the reference's reader returns them (kseq_records)."""
    with open(path, "rb") as f:
        head = f.read(2)
    op = gzip.open if head == b"\x1f\x8b" else open   # gzopen reads both (src/main.cpp:312)
    with op(path, "rb") as f:
        buf = f.read()
    return [(name.decode("latin-1"), np.frombuffer(seq, dtype=np.uint8)) for name, seq in kseq_records(buf)]


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
