#!/bin/bash
# calls in flight (engine slots x host threads of sa_seed_interval)
cd ${GRAFT_REPO_ROOT:-/root/repo}
: > gpurun_out/sweep_slots.txt
for n in 2 3 4; do
  BENCH_ARGS="--host-threads $n" bash tools/sweep_bench.sh "SEGALIGN_AMD_SLOTS=$n" >> gpurun_out/sweep_slots.txt 2>&1
done
