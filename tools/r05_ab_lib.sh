#!/bin/bash
# round 5: this tree against ab_exp/libsegalign_hip_old.so (the commit before, built here), tests first, then interleaved bench runs on one box
#   tools/r05_ab_lib.sh <tag> [tests to run first]
# The old library: git stash (or check out the other commit) && python -c 'from segalign_amd.build import build_lib; build_lib(force=True)' &&
#   mkdir -p ab_exp && cp segalign_amd/lib/libsegalign_hip.so ab_exp/libsegalign_hip_old.so, then come back and rebuild (ab_exp/ is git-ignored).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=$R/gpurun_out/${1:-r05ab}; mkdir -p $out; shift
timeout 1800 python -m pytest ${@:-tests/test_gpu_filter_audit.py tests/test_gpu_join.py tests/test_gpu_parity.py tests/test_gpu_lookup_paths.py tests/test_gpu_chain.py tests/test_gpu_random.py tests/test_gpu_edge_cases.py tests/test_gpu_rm_golden.py tests/test_gpu_rm_mask.py} -x -q > $out/tests.txt 2>&1; tail -4 $out/tests.txt
rm -rf /tmp/old; mkdir -p /tmp/old; cp -r $R/segalign_amd $R/bench.py $R/oracle $R/profiles $R/tests /tmp/old/ 2>/dev/null
cp $R/ab_exp/libsegalign_hip_old.so /tmp/old/segalign_amd/lib/libsegalign_hip.so
run() { name=$1; dir=$2; shift; shift
  (cd $dir; timeout 900 python bench.py --no-dropin --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err)
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
run new_$rep $R --steps 10 --warmup 3
run old_$rep /tmp/old --steps 10 --warmup 3
done
for w in lumpy notransition human rm; do
run ${w}_new $R --workload $w --steps 5 --warmup 2
run ${w}_old /tmp/old --workload $w --steps 5 --warmup 2
done
