#!/bin/bash
out=$PWD/gpurun_out/r04j; mkdir -p $out
b() { tree=$1; name=$2; shift 2; (cd $tree && python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err); python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  b ab/pre pre_default_$rep
  b . cur_default_$rep
  SEGALIGN_AMD_SLOTS=4 b . cur_default_slots4_$rep
done
for rep in 1 2; do
  SEGALIGN_AMD_CALL_HITS=0 b ab/pre pre_notrans20_$rep --workload notransition
  SEGALIGN_AMD_CALL_HITS=0 b . cur_notrans20_$rep --workload notransition
  b . cur_notrans_sized_$rep --workload notransition
  SEGALIGN_AMD_CALL_HITS=$((128<<20)) b . cur_notrans_128M_$rep --workload notransition
done
