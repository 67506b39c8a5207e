#!/bin/bash
out=$PWD/gpurun_out/r04w; mkdir -p $out
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); r=d["roofline"]; print("$name", d["value"], d["ms_per_step"], "cls single-stream us", r["single_stream"]["avg_launch_us"], "ss frac", r["single_stream"]["frac"], d["config"]["hsp_checksum"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  SEGALIGN_AMD_CLS_ONE_COPY=2 b nt_shifted_$rep --workload notransition
  SEGALIGN_AMD_CLS_ONE_COPY=1 b nt_onecopy_$rep --workload notransition
  SEGALIGN_AMD_CLS_ONE_COPY=2 b def_shifted_$rep
  SEGALIGN_AMD_CLS_ONE_COPY=1 b def_onecopy_$rep
done
SEGALIGN_AMD_CLS_ONE_COPY=2 b plumbing_shifted --workload plumbing --steps 50 --warmup 10
SEGALIGN_AMD_CLS_ONE_COPY=1 b plumbing_onecopy --workload plumbing --steps 50 --warmup 10
