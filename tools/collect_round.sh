#!/bin/bash
# Everything profiles/rNN holds, in one GPU-box run:  bash tools/collect_round.sh r03
set -u
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
SINGLE_STREAM=1 PMC_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1" \
  bash tools/profile_bench.sh $TAG --steps 3 --warmup 1 --no-dropin --host-threads 1 --intervals-in-flight 1 > gpurun_out/profile_$TAG.log 2>&1
python bench.py > gpurun_out/$TAG/bench_line_default.json 2> gpurun_out/$TAG/bench_line_default.err
for w in notransition rm human; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/$TAG/bench_line_$w.json 2> /dev/null
done
# (a plumbing step is ~1 ms: enough of them for a stable figure)
timeout 900 python bench.py --workload plumbing --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/$TAG/bench_line_plumbing.json 2> /dev/null
timeout 600 python tools/upload_overlap.py gpurun_out/$TAG/upload_overlap.txt > gpurun_out/upload_overlap.log 2>&1
timeout 600 python tools/timeline.py gpurun_out/$TAG/timeline.txt > gpurun_out/timeline.log 2>&1
SINGLE_STREAM=1 WORKLOAD=human TARGET_MBP=500.0 PMC_ARGS="--workload human --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1" \
  timeout 1500 bash tools/profile_bench.sh ${TAG}_human --workload human --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1 > gpurun_out/profile_${TAG}_human.log 2>&1
rm -f gpurun_out/$TAG/bench_under_pmc*.log gpurun_out/${TAG}_human/bench_under_pmc*.log
ls -la gpurun_out/$TAG gpurun_out/${TAG}_human
