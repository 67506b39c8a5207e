#!/usr/bin/env python3
"""HBM traffic per launch per kernel from the FETCH_SIZE / WRITE_SIZE PMC passes (prof_summary.py text output).

Units and gfx950 correction exactly as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes:
  FETCH_SIZE, WRITE_SIZE are in KiB-like units of 1024 B (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024);
  on gfx950 FETCH_SIZE counts 128-B fabric requests at 64 B, i.e. reads exactly 1/2 of the bytes fetched -> doubled.
  WRITE_SIZE is taken as reported (uncalibrated for narrow writes; noted in the output)."""
import json
import re
import sys


def parse(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+(sa::\S+.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = re.sub(r"<.*", "", m.group(1).replace("sa::", "")).strip()
            continue
        m = re.match(r"\s+%s\s+sum=(\S+)\s+per_dispatch=(\S+)" % counter, line)
        if m and cur:
            # several template instantiations share a base name: keep the largest (the un-instrumented hot one dominates)
            out[cur] = max(out.get(cur, 0.0), float(m.group(2)))
    return out


def main():
    f = parse(sys.argv[1], "FETCH_SIZE")
    w = parse(sys.argv[2], "WRITE_SIZE")
    res = {"_note": "bytes per launch; read = 2 * FETCH_SIZE * 1024 (gfx950 correction), write = WRITE_SIZE * 1024 "
                    "(uncalibrated for narrow stores)"}
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * f.get(k, 0.0) * 1024.0
        wr = w.get(k, 0.0) * 1024.0
        res[k] = {"read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes": round(rd + wr)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
