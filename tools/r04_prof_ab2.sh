#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04i; mkdir -p $out
for t in pre cur; do
  tree=$R/ab/$t; [ $t = cur ] && tree=$R
  rm -rf /tmp/raw_$t
  (cd $tree && SEGALIGN_AMD_CALL_HITS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw_$t -o r -- python bench.py --workload notransition --steps 4 --warmup 2 --no-roofline --no-cpu-baseline > $out/${t}_multi.log 2>&1)
  python $R/tools/prof_summary.py /tmp/raw_$t --out $out/${t}_multi_kernel_stats.txt
  # gaps: per stream idle analysis from the trace
  python - <<PY > $out/${t}_trace_summary.txt
import csv, glob, collections
f = glob.glob("/tmp/raw_$t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
ev = []
for r in rows: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = 0; depth = 0; last = t0; hist = collections.Counter()
for ts, d in ev:
    hist[min(depth, 8)] += ts - last; last = ts; depth += d
tot = t1 - t0
print("span ms", tot / 1e6, "kernels", len(rows))
for k in sorted(hist): print("concurrency", k, "share", round(hist[k] / tot, 4))
q = collections.Counter(r.get("Queue_Id") for r in rows)
print("queues", dict(q))
PY
done
