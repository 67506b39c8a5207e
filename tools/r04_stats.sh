#!/bin/bash
# single-stream kernel durations of one workload under rocprofv3 (top kernels):  tools/r04_stats.sh <tag> [bench args]
out=$PWD/gpurun_out/r04y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
v=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --host-threads 1 --intervals-in-flight 1 > $out/prof_$v.json 2> $out/prof_$v.err
f=$(find $out/prof_$v -name '*kernel_stats.csv' | head -1)
echo "== $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]: print("  %-60s %6s %12.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
rm -rf $out/prof_$v
