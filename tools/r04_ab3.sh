#!/bin/bash
out=$PWD/gpurun_out/r04e; mkdir -p $out
b() { name=$1; shift; python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"], d["config"]["calls_in_flight_per_gpu"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
 for s in 4 6 8; do
  SEGALIGN_AMD_SLOTS=$s b default_slots${s}_$rep
  SEGALIGN_AMD_SLOTS=$s SEGALIGN_AMD_CALL_HITS=0 b notrans20_slots${s}_$rep --workload notransition
  SEGALIGN_AMD_SLOTS=$s SEGALIGN_AMD_CALL_HITS=$((128<<20)) b notrans128M_slots${s}_$rep --workload notransition
 done
done
SEGALIGN_AMD_SLOTS=4 SEGALIGN_AMD_ARENA_VMM=0 b default_slots4_novmm
SEGALIGN_AMD_SLOTS=4 SEGALIGN_AMD_DEBUG=1 b default_slots4_debug; grep -i "granularity\|slot" $out/default_slots4_debug.err | head -5
