#!/usr/bin/env python3
"""Generator of tests/golden/kseq_golden.json: FASTA files with the corners a reader can stumble over, and the records the REFERENCE's own reader
returns for them -- oracle/_ref/kseq_dump: klib's kseq.h as vendored under /root/reference/common, compiled as it lies (no stand-in; zlib is in the
image) and instantiated / read as src/main.cpp:21,:318,:336-341 do.  A pin, not a second route: the host's FASTA readers (segalign_amd/fasta.py,
segalign_amd/host/host_common.hpp) are held against these vectors, and against the binary itself where it exists (tests/test_fasta_kseq.py).

usage: make -C oracle _ref && python tests/golden/make_kseq_golden.py   (needs /root/reference)
"""
import base64
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DUMP = os.path.join(ROOT, "oracle", "_ref", "kseq_dump")
OUT = os.path.join(HERE, "kseq_golden.json")


def files():
    long_line = b"ACGTTGCA" * 12000                      # one 96 000-byte sequence line (longer than the C++ reader's 64 KiB buffer)
    long_head = b">chrL " + b"x" * 70000                 # a header longer than that buffer
    yield "plain", b">chr1 first record\nACGTACGTAC\nGGGGCCCCAA\nTT\n>chr2\tsecond\nacgtnNRYKM\n", False
    yield "crlf", b">chr1 d\r\nACGT\r\nTTGA\r\n>chr2\r\nGG\r\n", False
    yield "no_final_newline", b">a\nACGT\n>b\nTTTT", False
    yield "empty_lines", b"\n\n>a x\n\nAC\n\nGT\n\n\n>b\n\nTT\n\n", False
    yield "empty_record", b">a\n>b\nACGT\n>c\n", False
    yield "leading_garbage", b"this is not fasta\nACGT\n>a\nTTTT\n", False
    yield "empty_name", b">\nACGT\n> desc only\nGG\n", False
    yield "spaces_in_sequence", b">a\nAC GT \n TTAA\n>b\nA\tC\n", False
    yield "gt_inside_line", b">a\nAC>GT\nTT\n", False
    yield "long_line_and_header", b">s\n" + long_line + b"\n" + long_head + b"\n" + long_line[:70001] + b"\nACGT\n", False
    yield "gzip", b">chr1 first\nACGTACGTAC\nGGGG\n>chr2\nacgtn\n", True
    yield "single_record_one_line", b">only\nACGTACGTNNNNacgt\n", False
    yield "fastq_like_lines", b">a\nACGT\n+\nIIII\n>b\nTT\n", False      # kseq is a FASTA / FASTQ reader: '+' opens a quality block
    yield "at_sign_line", b">a\nACGT\n@x\nTTTT\n>b\nGG\n", False          # ... and '@' a FASTQ record
    yield "fastq", b"@r1 d\nACGT\n+\nIIII\n@r2\nTTGG\nAA\n+r2\nIIII\nII\n>c\nGG\n", False
    yield "fastq_short_quality", b">a\nAC\n@r1\nACGT\n+\nII\n>b\nTT\n", False   # quality shorter than the sequence: kseq_read fails, the loop of main.cpp:336 ends
    yield "crlf_empty_line_after_header", b">a\r\n\r\nACGT\r\n\r\nTT\r\n>b\r\n\r\n", False   # a lone CR is a sequence character when it opens the sequence
    yield "header_only_at_eof", b">a\nACGT\n>", False
    yield "name_at_eof", b">a\nACGT\n>b", False
    yield "gt_in_leading_garbage", b"gar>bage\nACGT\n>a\nTT\n", False              # the first '>' ANYWHERE opens a header
    yield "single_cr_line", b">a\nA\r\nC\r\n", False
    yield "empty_file", b"", False
    yield "no_header", b"ACGT\nTTTT\n", False


def main():
    if not os.path.exists(DUMP):
        sys.exit("make -C oracle _ref first (needs /root/reference)")
    tmp = tempfile.mkdtemp(prefix="sa_kseq_")
    cases = []
    for name, data, gz in files():
        p = os.path.join(tmp, name + (".fa.gz" if gz else ".fa"))
        (gzip.open if gz else open)(p, "wb").write(data)
        out = subprocess.check_output([DUMP, p])
        recs = []
        for line in out.split(b"\n")[:-1]:
            nm, ln, seq = line.split(b"\t", 2)
            assert int(ln) == len(seq), (name, nm, ln, len(seq))
            recs.append([nm.decode(), base64.b64encode(seq).decode()])
        cases.append(dict(name=name, gz=gz, data=base64.b64encode(data).decode(), records=recs))
        print("%-24s %d records, %d bases" % (name, len(recs), sum(len(base64.b64decode(r[1])) for r in recs)))
    json.dump(dict(note="records (name, sequence; base64) the reference's own FASTA reader -- common/kseq.h compiled as it lies, oracle/_ref/kseq_dump -- returns for the "
                        "file `data` (base64; written through gzip where gz).  Generator: tests/golden/make_kseq_golden.py.", cases=cases), open(OUT, "w"))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
