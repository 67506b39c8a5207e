"""CPU: segalign_amd/fasta.py -- FASTA records -> '&'-joined sequence blocks the way the reference host lays them out
(src/main.cpp:336-409 for the query, :475-549 for the target): records are appended to the arena with ONE '&' between them,
and a block is closed as soon as its length -- separators included -- EXCEEDS the block size (the test runs after a record
was added, :359 / :515), so a block may be larger than the nominal size by up to one record."""
import gzip

import numpy as np
import pytest

from segalign_amd import fasta


def write_fasta(path, recs, width=60, gz=False, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    data = b""
    for name, seq in recs:
        data += b">" + name.encode() + b" some description" + nl
        for i in range(0, len(seq), width):
            data += seq[i:i + width] + nl
    (gzip.open if gz else open)(path, "wb").write(data)


RECS = [("chrI", b"ACGT" * 75), ("chrII", b"acgtnN" * 50), ("chrIII", b"TTTTGGGGCC" * 30), ("chrM", b"ACGTRYKM" * 5)]


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("crlf", [False, True])
def test_records_round_trip_plain_and_gzip(tmp_path, gz, crlf):
    p = str(tmp_path / ("g.fa.gz" if gz else "g.fa"))
    write_fasta(p, RECS, gz=gz, crlf=crlf)
    got = fasta.read_records(p)
    assert [n for n, _ in got] == [n for n, _ in RECS]           # names stop at the first blank (kseq name, main.cpp:338)
    assert [bytes(s) for _, s in got] == [s for _, s in RECS]    # line breaks removed, case and IUPAC letters kept


def reference_block_plan(lengths, block_size):
    """src/main.cpp:336-409 restated: seq_block_len += len; close if > size, else append '&' (+1)."""
    blocks, first, blen = [], 0, 0
    for i, n in enumerate(lengths):
        blen += n
        if blen > block_size:
            blocks.append((first, i + 1))
            first, blen = i + 1, 0
        else:
            blen += 1
    if first < len(lengths):
        blocks.append((first, len(lengths)))
    return blocks


def test_block_rule_closes_after_exceeding():
    assert fasta.plan_blocks([300, 300, 300], 500) == [(0, 2), (2, 3)]       # 300 + 1 + 300 = 601 > 500: closed AFTER the 2nd
    assert fasta.plan_blocks([600, 10], 500) == [(0, 1), (1, 2)]             # one record alone may exceed the size
    assert fasta.plan_blocks([250, 249], 500) == [(0, 2)]                    # 250 + 1 + 249 = 500 is not "greater than"
    assert fasta.plan_blocks([250, 250], 500) == [(0, 2)]                    # 501 > 500 closes it -- with both records inside
    rng = np.random.default_rng(1)
    for _ in range(200):
        lens = [int(x) for x in rng.integers(1, 400, int(rng.integers(1, 30)))]
        bs = int(rng.integers(50, 900))
        assert fasta.plan_blocks(lens, bs) == reference_block_plan(lens, bs), (lens, bs)


def test_first_block_layout(tmp_path):
    p = str(tmp_path / "g.fa")
    write_fasta(p, RECS)
    blk, nblocks = fasta.first_block(p, block_size=500)   # chrI (300) + '&' + chrII (300) = 601 > 500
    assert nblocks == 2
    assert bytes(blk) == RECS[0][1] + b"&" + RECS[1][1]    # one separator between records, none at the end of a block
    whole, n1 = fasta.first_block(p)                        # default 500 Mbp: everything in one block
    assert n1 == 1 and bytes(whole) == b"&".join(s for _, s in RECS)
    recs = fasta.read_records(p)
    assert bytes(fasta.block_bytes(recs, (2, 4))) == RECS[2][1] + b"&" + RECS[3][1]


def test_empty_and_headerless_input(tmp_path):
    p = str(tmp_path / "e.fa")
    open(p, "wb").write(b"ACGT\nACGT\n")                   # no header line: nothing is a record (kseq would read none either)
    assert fasta.read_records(p) == []
    open(p, "wb").write(b">only\n")
    got = fasta.read_records(p)
    assert len(got) == 1 and got[0][0] == "only" and got[0][1].size == 0
