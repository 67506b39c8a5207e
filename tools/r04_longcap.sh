#!/bin/bash
# second level: bases a hit may walk per side before it is handed to the exact stage as a candidate (option long_cap), same box
out=$PWD/gpurun_out/r04z; mkdir -p $out
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); r=d["roofline"]; k=r["kernels"]
    print("$name", d["value"], d["ms_per_step"], "cand_frac", r["per_hit"]["candidate_frac"], "surv", r["per_hit"]["survivor_frac"], d["config"]["hsp_checksum"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  for c in 64 128 192; do
    SEGALIGN_AMD_LONG_CAP=$c b nt_c${c}_$rep --workload notransition
    SEGALIGN_AMD_LONG_CAP=$c b def_c${c}_$rep
  done
done
SEGALIGN_AMD_LONG_CAP=64 b lumpy_c64 --workload lumpy
SEGALIGN_AMD_LONG_CAP=128 b lumpy_c128 --workload lumpy
