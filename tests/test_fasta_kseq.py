"""CPU: the host's two FASTA readers -- segalign_amd/fasta.py::kseq_records (bench.py's --target-fasta / --query-fasta) and
segalign_amd/host/host_common.hpp::read_fasta (the C++ hosts) -- against the REFERENCE's own reader: klib's kseq.h as vendored under
/root/reference/common, compiled as it lies (oracle/_ref/kseq_dump, instantiated and looped over as src/main.cpp:21,:318,:336 do; no stand-in,
zlib is in the image).  A pin: the committed vectors (tests/golden/kseq_golden.json: CRLF, empty lines, empty records and names, garbage in front
of the first header, FASTQ blocks, a lone CR opening a sequence, headers at the end of the file, lines and headers beyond 64 KiB, gzip) and, where
the binary exists (this container, the GPU box), a fuzz of random byte soups straight against it."""
import base64
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from segalign_amd import fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "kseq_golden.json")))["cases"]
KSEQ_DUMP = os.path.join(ROOT, "oracle", "_ref", "kseq_dump")


def parse_dump(out):
    recs, i = [], 0
    while i < len(out):
        t1 = out.index(b"\t", i)
        t2 = out.index(b"\t", t1 + 1)
        n = int(out[t1 + 1:t2])
        recs.append((out[i:t1], out[t2 + 1:t2 + 1 + n]))
        i = t2 + 1 + n + 1
    return recs


@pytest.fixture(scope="module")
def cpp_dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fasta") / "fasta_dump")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-w", "-I", os.path.join(ROOT, "segalign_amd", "host"), os.path.join(ROOT, "tests", "cpp", "fasta_dump.cpp"), "-o", exe, "-lz"])
    return exe


def want(c):
    return [(n.encode("latin-1"), base64.b64decode(s)) for n, s in c["records"]]


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_python_reader_returns_kseqs_records(c, tmp_path):
    assert fasta.kseq_records(base64.b64decode(c["data"])) == want(c)
    p = str(tmp_path / "f.fa")            # gzip is told by the file's magic, as gzopen tells it
    (gzip.open if c["gz"] else open)(p, "wb").write(base64.b64decode(c["data"]))
    assert [(n.encode("latin-1"), bytes(s)) for n, s in fasta.read_records(p)] == want(c)


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_cpp_reader_returns_kseqs_records(c, tmp_path, cpp_dump):
    p = str(tmp_path / "f.fa")
    (gzip.open if c["gz"] else open)(p, "wb").write(base64.b64decode(c["data"]))
    assert parse_dump(subprocess.check_output([cpp_dump, p])) == want(c)


def soups(seed, count):
    rng = np.random.default_rng(seed)
    alpha = [b"A", b"C", b"G", b"T", b"N", b">", b"@", b"+", b"\n", b"\n", b"\r\n", b"\r", b" ", b"\t", b"x"]
    for k, weights in enumerate(([8, 8, 8, 8, 2, 2, 1, 1, 5, 5, 2, 1, 2, 1, 2], [6, 6, 6, 6, 1, 3, 3, 4, 6, 6, 2, 1, 1, 1, 1])):
        w = np.array(weights, float) / sum(weights)
        for _ in range(count):
            n = int(rng.integers(0, 140))
            yield b"".join(alpha[i] for i in rng.choice(len(alpha), n, p=w))
    for _ in range(8):   # across the readers' buffer sizes (16 KiB in kseq, 64 KiB in the C++ reader): long lines, CRs at buffer ends
        body = b"".join(rng.choice([b"ACGT" * 4000 + b"\r\n", b"AC" * 9000 + b"\n", b"\r\n", b">h x\r\n", b"@q\n", b"+\n"], int(rng.integers(3, 12))).tolist())
        yield body


@pytest.mark.skipif(not os.path.exists(KSEQ_DUMP), reason="oracle/_ref/kseq_dump is built where /root/reference exists (and travels to the GPU box)")
def test_both_readers_against_the_reference_reader_itself(tmp_path, cpp_dump):
    p = str(tmp_path / "s.fa")
    for c in CASES:     # the committed vectors are what the binary says today
        (gzip.open if c["gz"] else open)(p, "wb").write(base64.b64decode(c["data"]))
        assert parse_dump(subprocess.check_output([KSEQ_DUMP, p])) == want(c), c["name"]
    n = 0
    for data in soups(11, 1500):
        open(p, "wb").write(data)
        ref = parse_dump(subprocess.check_output([KSEQ_DUMP, p]))
        assert fasta.kseq_records(data) == ref, data[:200]
        assert parse_dump(subprocess.check_output([cpp_dump, p])) == ref, data[:200]
        n += 1
    assert n > 3000
