#!/bin/bash
# round 5: key-ordered calls against streamed calls, whole pass, one box: call size, workloads
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05k}; mkdir -p $out
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-dropin --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-28s value %.4f ms %.2f calls %s cpc %s ss_filter_us %s" % ("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"], d["config"]["chunks_per_call"], r["single_stream"]["avg_launch_us"]))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
run ko0_$rep SEGALIGN_AMD_KEY_ORDER=0 -- --steps 10 --warmup 3
run ko1_200_$rep SEGALIGN_AMD_KEY_ORDER=1 -- --steps 10 --warmup 3
run ko1_100_$rep SEGALIGN_AMD_KEY_ORDER=1 SEGALIGN_AMD_KEY_ORDER_CHUNKS=100 -- --steps 10 --warmup 3
run ko1_134_$rep SEGALIGN_AMD_KEY_ORDER=1 SEGALIGN_AMD_KEY_ORDER_CHUNKS=134 -- --steps 10 --warmup 3
done
run lumpy_ko0 SEGALIGN_AMD_KEY_ORDER=0 -- --workload lumpy --steps 4 --warmup 1
run lumpy_ko1 SEGALIGN_AMD_KEY_ORDER=1 -- --workload lumpy --steps 4 --warmup 1
run lumpy_ko1_100 SEGALIGN_AMD_KEY_ORDER=1 SEGALIGN_AMD_KEY_ORDER_CHUNKS=100 -- --workload lumpy --steps 4 --warmup 1
run notr_ko0 SEGALIGN_AMD_KEY_ORDER=0 -- --workload notransition --steps 10 --warmup 3
run notr_ko2 SEGALIGN_AMD_KEY_ORDER=2 -- --workload notransition --steps 10 --warmup 3
run rm_ko0 SEGALIGN_AMD_KEY_ORDER=0 -- --workload rm --steps 5 --warmup 2
run rm_ko1 SEGALIGN_AMD_KEY_ORDER=1 -- --workload rm --steps 5 --warmup 2
