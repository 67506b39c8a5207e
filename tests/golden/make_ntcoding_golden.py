#!/usr/bin/env python3
"""Generate tests/golden/ntcoding_golden.json from the REAL reference object oracle/_ref/libntcoding_ref.so
(common/ntcoding.cpp compiled as it lies under /root/reference by `make -C oracle _ref`).
Runs only in the authoring container; the JSON (inputs + expected outputs, no reference text) is committed."""
import ctypes as C
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    O.build(with_ref=True)
    R = O.ref_lib()
    assert R is not None, "reference object not built (need /root/reference)"
    rnd = random.Random(20260928)
    shapes = ["TTT0T00TT00T0T0TTTT", "TTT0T0TT00TT00T0T0TTTT", "1110100110010101111", "TTTTTTTTTTTT", "T0T0T0T0T", "1T1T0001T"]
    alphabet = "ACGT" * 12 + "acgtNn&XRY-"
    cases = []
    for shape in shapes:
        k = R.ref_GenerateShapePos(shape.encode())
        trans = [R.ref_IsTransitionAtPos(t) for t in range(k)]
        seqs = []
        for _ in range(6):
            n = rnd.randint(len(shape), 90)
            seq = "".join(rnd.choice(alphabet) for _ in range(n))
            kmers = [R.ref_GetKmerIndexAtPos(seq.encode(), p, len(shape)) for p in range(n - len(shape) + 1)]
            seqs.append({"seq": seq, "kmers": kmers})
        clean = "".join(rnd.choice("ACGT") for _ in range(200))
        kmers = [R.ref_GetKmerIndexAtPos(clean.encode(), p, len(shape)) for p in range(200 - len(shape) + 1)]
        seqs.append({"seq": clean, "kmers": kmers})
        cases.append({"shape": shape, "kmer_size": k, "transition": trans, "seqs": seqs})
    rc = []
    for _ in range(8):
        n = rnd.randint(1, 120)
        seq = "".join(rnd.choice("ACGTacgtNn&") for _ in range(n))
        start = rnd.randint(0, n - 1)
        ln = rnd.randint(1, n - start)
        dst = C.create_string_buffer(ln + 4)
        R.ref_RevComp(dst, seq.encode(), 0, start, ln)
        rc.append({"seq": seq, "start": start, "len": ln, "rc": dst.raw[:ln].decode()})
    out = {"source": "common/ntcoding.cpp via oracle/_ref/libntcoding_ref.so", "kmer_cases": cases, "revcomp_cases": rc}
    with open(os.path.join(HERE, "ntcoding_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(cases), "shape cases,", len(rc), "revcomp cases")


if __name__ == "__main__":
    main()
