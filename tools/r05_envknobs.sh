#!/bin/bash
# round 5: runtime environment knobs on the default pass, one box, interleaved (bench.py --steps 10 --warmup 3 --no-dropin --no-cpu-baseline --no-roofline)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05env}; mkdir -p $out
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-dropin --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); print("%-26s %.4f Gbp/s  %.2f ms" % ("$name", d["value"], d["ms_per_step"]))
except Exception as e:
    print("$name failed", e); print(open("$out/$name.err").read()[-600:])
PY
}
for rep in 1 2 3; do
run base_$rep A=1 --
run devkernarg_$rep HIP_FORCE_DEV_KERNARG=1 --
run hwq12_$rep GPU_MAX_HW_QUEUES=12 --
run hwq16_$rep GPU_MAX_HW_QUEUES=16 --
run nosdma_$rep HSA_ENABLE_SDMA=0 --
done
