#!/bin/bash
out=$PWD/gpurun_out/r04p; mkdir -p $out
b() { name=$1; shift; python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"], d["config"]["calls_in_flight_per_gpu"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  b d40_$rep
  b d40_ht2_$rep --host-threads 2
  b d40_ht4_$rep --host-threads 4
  b d80_$rep --chunks-per-call 80
  b d60_$rep --chunks-per-call 60
  b d20_$rep --chunks-per-call 20
  SEGALIGN_AMD_SLOTS=4 b d40_slots4_$rep
done
b nt --workload notransition
b lumpy --workload lumpy --steps 3 --warmup 1
b human --workload human --steps 2 --warmup 1
b rm --workload rm --steps 2 --warmup 1
b plumbing --workload plumbing --steps 50 --warmup 10
