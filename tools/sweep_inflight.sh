#!/bin/bash
# calls in flight: host threads per interval x intervals in flight
cd ${GRAFT_REPO_ROOT:-/root/repo}
: > gpurun_out/sweep_inflight.txt
for cfg in "2 1" "3 1" "2 2" "3 2" "2 3" "1 3" "1 4"; do
  set -- $cfg
  BENCH_ARGS="--host-threads $1 --intervals-in-flight $2" bash tools/sweep_bench.sh "X=ht$1_iv$2" >> gpurun_out/sweep_inflight.txt 2>&1
done
