#!/usr/bin/env python3
"""End-to-end run of the C++ host harness on the BASELINE configs[1] stand-in (writes FASTA, runs segalign_host --debug).
Run on the GPU box: python tools/run_host_fullsize.py [target_mbp] ; prints the harness's own timing lines."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segalign_amd import synth  # noqa: E402
from segalign_amd.build import build_host  # noqa: E402


def write_fasta(path, name_prefix, joined):
    recs = bytes(joined).split(b"&")
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">%s%d\n" % (name_prefix, i + 1))
            for j in range(0, len(r), 1 << 20):
                f.write(r[j:j + (1 << 20)] + b"\n")


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    threads = sys.argv[2] if len(sys.argv) > 2 else "4"
    t, q = synth.make_pair(int(mbp * 1e6), 3, 4, sub_rate=0.08, mask_frac=0.2, records=7, invert_frac=0.3, invert_block=100_000)
    d = tempfile.mkdtemp(prefix="sa_host_")
    write_fasta(os.path.join(d, "t.fa"), b"chrT", t)
    write_fasta(os.path.join(d, "q.fa"), b"chrQ", q)
    out = os.path.join(d, "out")
    os.mkdir(out)
    exe = build_host()
    t0 = time.time()
    res = subprocess.run([exe, os.path.join(d, "t.fa"), os.path.join(d, "q.fa"), "./", "--outdir=" + out, "--debug",
                          "--num_threads=" + threads, "--num_gpu=1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.time() - t0
    err = res.stderr.decode().split("\n")
    print("\n".join(l for l in err if l.startswith("Time elapsed") or l.startswith("#")))
    nseg = sum(1 for f in os.listdir(out) if f.endswith(".segments"))
    nlines = sum(sum(1 for _ in open(os.path.join(out, f))) for f in os.listdir(out) if f.endswith(".segments"))
    print("exit=%d wall=%.2fs segment_files=%d hsp_lines=%d lastz_cmds=%d" % (res.returncode, wall, nseg, nlines,
                                                                             len(res.stdout.decode().strip().split("\n"))))


if __name__ == "__main__":
    main()
