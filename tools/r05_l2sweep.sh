#!/bin/bash
# round 5: second level launch geometry re-swept after the context cut (a third of the records it used to get)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05l2}; mkdir -p $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-dropin --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-22s value %.4f ms %.2f l2_ss %s filter_ss %.0f" % ("$name", d["value"], d["ms_per_step"], r["profile_check"]["scopes"].get("extend_filter2",{}).get("events_us"), r["single_stream"]["avg_launch_us"]))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
for b in 512 128 256 1024 2048; do run l2b${b}_$rep SEGALIGN_AMD_L2_BLOCKS=$b -- --steps 10 --warmup 3; done
run fin32_$rep SEGALIGN_AMD_FIN_BATCH=32 -- --steps 10 --warmup 3
run fin56_$rep SEGALIGN_AMD_FIN_BATCH=56 -- --steps 10 --warmup 3
done
