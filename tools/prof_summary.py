#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace stats and/or PMC counter collection) per kernel.

usage: prof_summary.py <rocprofv3 output dir> [--out summary.txt]
Reads every *_kernel_stats.csv / *_kernel_trace.csv / *_counter_collection.csv under the directory and prints
per kernel: launches, total/avg duration and the SUM of each counter over all dispatches of that kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:70]


def main():
    d = sys.argv[1]
    out = sys.stdout
    if "--out" in sys.argv:
        out = open(sys.argv[sys.argv.index("--out") + 1], "w")
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        out.write("== %s\n" % os.path.relpath(f, d))
        for i, row in enumerate(csv.DictReader(open(f))):
            if i >= 40:
                break
            out.write("  %-70s calls=%s total_ns=%s avg_ns=%s pct=%s\n" % (
                short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
                row.get("Percentage")))
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0])
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        out.write("== %s (from trace)\n" % os.path.relpath(f, d))
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            out.write("  %-70s calls=%d total_us=%.1f avg_us=%.2f\n" % (k, n, t / 1e3, t / 1e3 / n))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        meta = {}
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
            meta[k] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Workgroup_Size"))
        out.write("== %s\n" % os.path.relpath(f, d))
        for k in sorted(agg, key=lambda k: -len(calls[k])):
            out.write("  %-70s dispatches=%d vgpr=%s sgpr=%s lds=%s wg=%s\n" % ((k, len(calls[k])) + meta[k]))
            for c, v in sorted(agg[k].items()):
                out.write("      %-28s sum=%.6g  per_dispatch=%.6g\n" % (c, v, v / max(len(calls[k]), 1)))


if __name__ == "__main__":
    main()
