#!/bin/bash
out=$PWD/gpurun_out/r04k; mkdir -p $out
b() { tree=$1; name=$2; shift 2; (cd $tree && python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err); python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  for t in pre p1 p2 p3 p4 cur; do tree=ab/$t; [ $t = cur ] && tree=.; b $tree ${t}_default_$rep; done
  SEGALIGN_AMD_CHAIN_CAP=4194304 b . cur_chaincap4M_default_$rep
done
for rep in 1 2; do
  b . cur_notrans_sized_$rep --workload notransition
  SEGALIGN_AMD_CALL_HITS=$((128<<20)) b . cur_notrans_128M_$rep --workload notransition
done
b . cur_lumpy --workload lumpy --steps 3 --warmup 1
