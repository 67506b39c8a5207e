#!/bin/bash
# round 5, after the seed-bound change: call size and calls in flight re-swept on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05w}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_filter_audit.py -x -q > $out/tests.txt 2>&1; tail -4 $out/tests.txt
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-dropin --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
    print("%-22s value %.4f ms %.2f calls %s" % ("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"]))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
run base_$rep X=1 -- --steps 10 --warmup 3
for k in 20 50 67 80 100; do run cpc${k}_$rep X=1 -- --steps 10 --warmup 3 --chunks-per-call $k; done
run fly8_$rep SEGALIGN_AMD_SLOTS=8 -- --steps 10 --warmup 3 --host-threads 4 --intervals-in-flight 2
run fly4_$rep X=1 -- --steps 10 --warmup 3 --host-threads 2 --intervals-in-flight 2
run fly5_$rep X=1 -- --steps 10 --warmup 3 --host-threads 5 --intervals-in-flight 1
done
