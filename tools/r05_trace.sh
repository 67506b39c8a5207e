#!/bin/bash
# round 5: per-kernel times of one pass with ONE call in flight, key-ordered calls on and off (same box)
#   tools/r05_trace.sh <tag> [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05t}; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$TAG; mkdir -p $out
cd $R
for ko in ${KOS:-1 0}; do
rm -rf /tmp/raw_$ko
SEGALIGN_AMD_KEY_ORDER=$ko timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw_$ko -o r -- python bench.py --steps 3 --warmup 1 --no-dropin --no-cpu-baseline --host-threads 1 --intervals-in-flight 1 "$@" > $out/bench_ko$ko.log 2>&1
python tools/prof_summary.py /tmp/raw_$ko --out $out/kernel_stats_ko$ko.txt
grep '^{"metric"' $out/bench_ko$ko.log | tail -1 > $out/bench_line_ko$ko.json
rm -rf /tmp/raw_$ko
done
head -45 $out/kernel_stats_ko1.txt
