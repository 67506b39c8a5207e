#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05sc}; mkdir -p $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-dropin --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-22s value %.4f ms %.2f fwd %.4f filter_ss_us %.0f l2_ss %s" % ("$name", d["value"], d["ms_per_step"], r["per_hit"]["forwarded_frac"], r["single_stream"]["avg_launch_us"], r["profile_check"]["scopes"].get("extend_filter2",{}).get("events_us")))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-1500:])
PY
}
for rep in 1 2 3; do
run notr_skip1_$rep SEGALIGN_AMD_CTX_SKIP_SEED=1 -- --workload notransition --steps 20 --warmup 5
run notr_skip0_$rep SEGALIGN_AMD_CTX_SKIP_SEED=0 -- --workload notransition --steps 20 --warmup 5
done
for rep in 1 2; do
run lumpy_skip1_$rep SEGALIGN_AMD_CTX_SKIP_SEED=1 -- --workload lumpy --steps 4 --warmup 1
run lumpy_skip0_$rep SEGALIGN_AMD_CTX_SKIP_SEED=0 -- --workload lumpy --steps 4 --warmup 1
done
run plumb_skip1 SEGALIGN_AMD_CTX_SKIP_SEED=1 -- --workload plumbing --steps 50 --warmup 10
run plumb_skip0 SEGALIGN_AMD_CTX_SKIP_SEED=0 -- --workload plumbing --steps 50 --warmup 10
