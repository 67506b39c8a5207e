#!/usr/bin/env python3
"""End-to-end run of the repeat-masker host harness (BASELINE configs[3] stand-in: self-alignment of a repeat-rich
synthetic genome).  Run on the GPU box: python tools/run_rm_host_fullsize.py [mbp] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from segalign_amd import synth  # noqa: E402
from segalign_amd.build import build_host, RM_HOST_BIN  # noqa: E402


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    threads = sys.argv[2] if len(sys.argv) > 2 else "4"
    n = int(mbp * 1e6)
    rng = np.random.default_rng(17)
    g = synth.random_dna(n, 5)
    fams = [synth.random_dna(int(rng.integers(300, 3000)), 100 + f) for f in range(40)]  # 40 repeat families
    for k in range(n // 20000):  # one copy every ~20 kb, 3-15 % diverged, both orientations
        u = fams[int(rng.integers(0, len(fams)))]
        cp = synth.mutate(u, 1000 + k, float(rng.uniform(0.03, 0.15)))
        p = int(rng.integers(0, n - cp.size))
        g[p:p + cp.size] = cp if k % 2 else synth.reverse_complement(cp)
    g = synth.soft_mask(g, 6, 0.1, 200, 2000)
    d = tempfile.mkdtemp(prefix="sa_rm_")
    with open(os.path.join(d, "g.fa"), "wb") as f:
        per = n // 6
        for i in range(6):
            f.write(b">chr%d\n" % (i + 1))
            r = bytes(g[i * per:(i + 1) * per])
            for j in range(0, len(r), 1 << 20):
                f.write(r[j:j + (1 << 20)] + b"\n")
    out = os.path.join(d, "out")
    os.mkdir(out)
    build_host()
    t0 = time.time()
    res = subprocess.run([RM_HOST_BIN, os.path.join(d, "g.fa"), "--outdir=" + out, "--debug", "--num_threads=" + threads,
                          "--num_gpu=1"] + sys.argv[3:], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.time() - t0
    err = res.stderr.decode().split("\n")
    print("\n".join(l for l in err if l.startswith("Time elapsed") or l.startswith("#")))
    files = [f for f in os.listdir(out) if f.endswith(".intervals")]
    nlines = sum(sum(1 for _ in open(os.path.join(out, f))) for f in files)
    print("exit=%d wall=%.2fs interval_files=%d intervals=%d" % (res.returncode, wall, len(files), nlines))
    if res.returncode:
        print("\n".join(err[-15:]))


if __name__ == "__main__":
    main()
