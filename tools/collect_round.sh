#!/bin/bash
# Everything profiles/rNN holds, in one GPU-box run:  bash tools/collect_round.sh r03
# Order: the rocprofv3 collections first (default workload, human-scale block pair), their summaries copied into profiles/ of the
# box's repo copy, THEN the bench lines -- so that the lines quote traffic / counters of the collection made beside them.
set -u
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out
SINGLE_STREAM=1 PMC_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1" \
  timeout 1200 bash tools/profile_bench.sh $TAG --steps 3 --warmup 1 --no-dropin --host-threads 1 --intervals-in-flight 1 > gpurun_out/profile_$TAG.log 2>&1 < /dev/null
SINGLE_STREAM=1 WORKLOAD=human TARGET_MBP=500.0 PMC_ARGS="--workload human --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1" \
  timeout 1500 bash tools/profile_bench.sh ${TAG}_human --workload human --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1 > gpurun_out/profile_${TAG}_human.log 2>&1 < /dev/null
rm -f gpurun_out/$TAG/bench_under_pmc*.log gpurun_out/${TAG}_human/bench_under_pmc*.log
for t in $TAG ${TAG}_human; do
  mkdir -p profiles/$t
  cp gpurun_out/$t/kernel_stats.txt gpurun_out/$t/pmc*.txt gpurun_out/$t/traffic.json gpurun_out/$t/workload.json gpurun_out/$t/commands.txt profiles/$t/ 2> /dev/null
done
timeout 600 python bench.py > gpurun_out/$TAG/bench_line_default.json 2> gpurun_out/$TAG/bench_line_default.err < /dev/null
for w in notransition rm human; do
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/$TAG/bench_line_$w.json 2> /dev/null < /dev/null
done
# (a plumbing step is ~1 ms: enough of them for a stable figure)
timeout 900 python bench.py --workload plumbing --steps 50 --warmup 10 > gpurun_out/$TAG/bench_line_plumbing.json 2> /dev/null < /dev/null
timeout 600 python tools/upload_overlap.py gpurun_out/$TAG/upload_overlap.txt > gpurun_out/upload_overlap.log 2>&1 < /dev/null
timeout 600 python tools/timeline.py gpurun_out/$TAG/timeline.txt > gpurun_out/timeline.log 2>&1 < /dev/null
ls -la gpurun_out/$TAG gpurun_out/${TAG}_human
