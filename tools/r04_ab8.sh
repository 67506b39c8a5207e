#!/bin/bash
out=$PWD/gpurun_out/r04z; mkdir -p $out
b() { name=$1; shift; python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  b nt_$rep --workload notransition
  SEGALIGN_AMD_CHAIN_SORT_THREADS=512 b nt_sort512_$rep --workload notransition
  SEGALIGN_AMD_CHAIN_SORT_THREADS=128 b nt_sort128_$rep --workload notransition
  b def_$rep
  SEGALIGN_AMD_CHAIN_SORT_THREADS=512 b def_sort512_$rep
  (cd ab/noagg && SEGALIGN_AMD_CLS_ONE_COPY=2 python bench.py --no-roofline --no-cpu-baseline --steps 8 --warmup 3 --workload notransition 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('nt_round_start_equiv', d['value'], d['ms_per_step'])")
done
