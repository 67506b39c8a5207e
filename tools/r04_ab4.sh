#!/bin/bash
out=$PWD/gpurun_out/r04h; mkdir -p $out
b() { tree=$1; name=$2; shift 2; (cd $tree && python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err); python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  for t in pre cur va vb vc vd; do
    tree=ab/$t; [ $t = cur ] && tree=.
    SEGALIGN_AMD_CALL_HITS=0 b $tree ${t}_notrans20_$rep --workload notransition
  done
done
for t in pre cur vc vd; do tree=ab/$t; [ $t = cur ] && tree=.; b $tree ${t}_default; done
