#!/usr/bin/env python3
"""Does a query-block upload overlap SeedAndFilter kernels that run on the other device buffer?  (SURVEY f-3)

Run ON THE GPU BOX:  python tools/upload_overlap.py [out.txt]
Generates a 60 Mbp target and a query of 4 blocks of ~50 Mbp (--seq_block_size=40000000), runs the C++ host harness
(segalign_amd/host/segalign_host.cpp: the next block is uploaded by a background thread into the other BUFFER_DEPTH slot,
src/main.cpp:649-685) under `rocprofv3 --kernel-trace --memory-copy-trace` (no counters), and reports for every
host-to-device copy >= 8 MiB how much of its duration lies inside kernel executions of the SAME process -- i.e. real
copy / compute overlap.  Only the small text summary is kept."""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from segalign_amd import synth  # noqa: E402
from segalign_amd.build import build_host  # noqa: E402


def write_fasta(path, recs):
    with open(path, "wb") as f:
        for name, s in recs:
            f.write(b">" + name.encode() + b"\n")
            b = bytes(s)
            for j in range(0, len(b), 1 << 20):
                f.write(b[j:j + (1 << 20)] + b"\n")


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    d = tempfile.mkdtemp(prefix="sa_overlap_")
    t = synth.random_dna(60_000_000, 11)
    q_recs = []
    for i in range(4):
        piece = synth.mutate(t[i * 12_000_000:i * 12_000_000 + 50_000_000 - i * 9_000_000], 50 + i, 0.07)
        q_recs.append(("q%d" % i, synth.soft_mask(piece, 60 + i, 0.15)))
    write_fasta(os.path.join(d, "t.fa"), [("t0", synth.soft_mask(t, 12, 0.15))])
    write_fasta(os.path.join(d, "q.fa"), q_recs)
    os.mkdir(os.path.join(d, "out"))
    exe = build_host()
    raw = os.path.join(d, "raw")
    cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", raw, "-o", "r", "--", exe,
           os.path.join(d, "t.fa"), os.path.join(d, "q.fa"), "./", "--seq_block_size=40000000", "--outdir=" + os.path.join(d, "out"),
           "--num_threads=4", "--num_gpu=1", "--nogapped", "--debug"]
    env = dict(os.environ, TMPDIR="/tmp")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp", env=env)
    out.write("command: %s\nexit: %d\n" % (" ".join(cmd[:8] + ["segalign_host", "t.fa(60 Mbp)", "q.fa(4 blocks)"] + cmd[13:]), res.returncode))
    for line in res.stderr.decode().split("\n"):
        if line.startswith("Time elapsed") or line.startswith("#"):
            out.write("  " + line + "\n")
    if res.returncode != 0:
        out.write("stderr tail:\n" + "\n".join(res.stderr.decode(errors="replace").split("\n")[-25:]) + "\n")
    kern = []
    for f in glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kern.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    kern.sort()
    copies = []
    for f in glob.glob(os.path.join(raw, "**", "*memory_copy_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            direction = row.get("Direction", "") or row.get("Kind", "")
            if "HOST_TO_DEVICE" not in direction.upper().replace(" ", "_") and "H2D" not in direction.upper():
                continue
            copies.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), int(row.get("Bytes", row.get("Size", 0)) or 0)))
    big = [c for c in copies if c[2] >= (8 << 20) or (c[2] == 0 and c[1] - c[0] > 200_000)]
    out.write("kernels traced: %d ; host-to-device copies: %d (%d of >= 8 MiB)\n" % (len(kern), len(copies), len(big)))
    if not kern or not big:
        out.write("nothing to compare\n")
        return
    t_first, t_last = kern[0][0], max(k[1] for k in kern)
    # skip copies before the first SeedAndFilter kernel (target + first query block: nothing to overlap with yet)
    first_saf = min((k[0] for k in kern if "extend_filter" in k[2]), default=t_first)
    tot = ov = 0
    rows = []
    for (s, e, n) in big:
        if e <= first_saf:
            continue
        o = 0
        for (ks, ke, name) in kern:
            if ke <= s:
                continue
            if ks >= e:
                break
            if any(x in name for x in ("encode", "pack", "rev_comp", "row_code")):  # the upload's own kernels do not count
                continue
            o += max(0, min(e, ke) - max(s, ks))
        o = min(o, e - s)
        rows.append((s - t_first, e - s, n, o))
        tot += e - s
        ov += o
    out.write("copies issued while SeedAndFilter is running (t = ns after the first kernel):\n")
    for (t0, dur, n, o) in rows[:40]:
        out.write("  t=%12d  dur=%9d ns  bytes=%10d  inside compute kernels: %5.1f %%\n" % (t0, dur, n, 100.0 * o / max(dur, 1)))
    out.write("total: %.2f ms of upload DMA, %.2f ms (%.1f %%) of it concurrent with compute kernels\n" %
              (tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1)))


if __name__ == "__main__":
    main()
