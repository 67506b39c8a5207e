#!/usr/bin/env python3
"""How full is the GPU?  Run ON THE GPU BOX:  python tools/timeline.py [out.txt] [bench args...]
Runs bench.py under `rocprofv3 --kernel-trace` (no counters) and reduces the trace of the steady state (the last 60 % of the
kernel time span) to: wall time, time with no kernel running, time with exactly / at least one context-filter kernel running,
time with two of them overlapping, and the same for the second level -- the figures behind DESIGN.md 10 item 1."""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap2(iv):
    """time during which at least two intervals of the list are open"""
    ev = []
    for s, e in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, tot = 0, None, 0
    for t, d in ev:
        if depth >= 2:
            tot += t - last
        depth += d
        last = t
    return tot


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    extra = sys.argv[2:] or ["--steps", "4", "--warmup", "1", "--no-roofline"]
    d = tempfile.mkdtemp(prefix="sa_timeline_")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--", sys.executable, os.path.join(ROOT, "bench.py")] + extra
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if not rows:
        out.write("no kernels traced (exit %d)\n%s\n" % (res.returncode, res.stderr.decode()[-800:]))
        return
    l1_all = [(s, e) for s, e, n in rows if "extend_filter_cls_kernel" in n]
    # steady state: from 30 % to 95 % of the level-1 filter launches (skips setup and warmup; the run has no extra passes)
    a, b = l1_all[int(len(l1_all) * 0.30)][0], l1_all[int(len(l1_all) * 0.95)][0]
    win = [(max(s, a), min(e, b), n) for s, e, n in rows if e > a and s < b]
    wall = b - a
    allk = [(s, e) for s, e, n in win]
    l1 = [(s, e) for s, e, n in win if "extend_filter_cls_kernel" in n]
    l2 = [(s, e) for s, e, n in win if "extend_filter_packed_kernel" in n]
    small = [(s, e) for s, e, n in win if "extend_filter_cls_kernel" not in n and "extend_filter_packed_kernel" not in n]
    out.write("command: bench.py %s\n" % " ".join(extra))
    out.write("window: %.1f ms, %d kernels, %d context-filter launches\n" % (wall / 1e6, len(win), len(l1)))
    def pct(x):
        return "%7.2f ms = %5.1f %%" % (x / 1e6, 100.0 * x / wall)
    out.write("no kernel running              : %s\n" % pct(wall - union(allk)))
    out.write("context filter running (>= 1)  : %s\n" % pct(union(l1)))
    out.write("two context filters overlapping: %s\n" % pct(overlap2(l1)))
    out.write("second level running           : %s\n" % pct(union(l2)))
    out.write("only small kernels running     : %s\n" % pct(union(allk) - union(l1 + l2)))
    out.write("sum of context-filter durations: %s  (avg %.1f us)\n" % (pct(sum(e - s for s, e in l1)), sum(e - s for s, e in l1) / max(len(l1), 1) / 1e3))
    out.write("sum of small-kernel durations  : %s\n" % pct(sum(e - s for s, e in small)))


if __name__ == "__main__":
    main()
