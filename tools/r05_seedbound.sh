#!/bin/bash
# round 5: context records in front of the seed (option ctx_skip_seed): soundness tests, then the same-box A/B of the default pass
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05sb}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_filter_audit.py tests/test_gpu_join.py tests/test_gpu_block_edges.py tests/test_gpu_parity.py tests/test_gpu_lookup_paths.py tests/test_gpu_random.py tests/test_gpu_edge_cases.py tests/test_gpu_find_hsps_golden.py tests/test_gpu_rm_golden.py -x -q > $out/tests.txt 2>&1; tail -6 $out/tests.txt
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-dropin --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-22s value %.4f ms %.2f fwd %.4f cand %.5f filter_ss_us %.0f l2_ss %s hsps %d chk %s" % ("$name", d["value"], d["ms_per_step"], r["per_hit"]["forwarded_frac"], r["per_hit"]["candidate_frac"], r["single_stream"]["avg_launch_us"], r["profile_check"]["scopes"].get("extend_filter2",{}).get("events_us"), d["config"]["hsps_per_step"], d["config"]["hsp_checksum"]))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-1500:])
PY
}
for rep in 1 2 3; do
run skip1_$rep SEGALIGN_AMD_CTX_SKIP_SEED=1 -- --steps 10 --warmup 3
run skip0_$rep SEGALIGN_AMD_CTX_SKIP_SEED=0 -- --steps 10 --warmup 3
done
for w in lumpy notransition human rm; do
run ${w}_skip1 SEGALIGN_AMD_CTX_SKIP_SEED=1 -- --workload $w --steps 3 --warmup 1
run ${w}_skip0 SEGALIGN_AMD_CTX_SKIP_SEED=0 -- --workload $w --steps 3 --warmup 1
done
