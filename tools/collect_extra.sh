#!/bin/bash
# rocprofv3 collections for the other bench workloads (notransition, rm, lumpy), so that their bench lines quote measured traffic:
#   bash tools/collect_extra.sh r03      -> gpurun_out/r03_notransition, gpurun_out/r03_rm (+ copies under profiles/ of the box)
set -u
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for w in ${WORKLOADS:-notransition rm lumpy}; do
  SINGLE_STREAM=1 WORKLOAD=$w PMC_ARGS="--workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1" \
    timeout 1200 bash tools/profile_bench.sh ${TAG}_$w --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1 > gpurun_out/profile_${TAG}_$w.log 2>&1 < /dev/null
  rm -f gpurun_out/${TAG}_$w/bench_under_pmc*.log
  mkdir -p profiles/${TAG}_$w
  cp gpurun_out/${TAG}_$w/kernel_stats.txt gpurun_out/${TAG}_$w/pmc*.txt gpurun_out/${TAG}_$w/traffic.json gpurun_out/${TAG}_$w/workload.json gpurun_out/${TAG}_$w/commands.txt profiles/${TAG}_$w/ 2> /dev/null
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/${TAG}_$w/bench_line_$w.json 2> /dev/null < /dev/null
done
ls -la gpurun_out/${TAG}_*
