#!/bin/bash
# class filter launch geometry after the one-copy query windows: workgroup size and wave budget, default + sparse-hit workload
out=$PWD/gpurun_out/r04s; mkdir -p $out
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --steps 5 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); r=d["roofline"]
    print("$name", d["value"], d["ms_per_step"], "cls single-stream us", r["single_stream"]["avg_launch_us"])
except Exception as e: print("$name FAILED", e)
PY
}
for t in 0 256 512 768 1024; do
  SEGALIGN_AMD_CTX_THREADS=$t b def_t$t
  SEGALIGN_AMD_CTX_THREADS=$t b nt_t$t --workload notransition
done
for w in 4096 8192 16384 32768; do
  SEGALIGN_AMD_CTX_WAVES=$w b def_w$w
  SEGALIGN_AMD_CTX_WAVES=$w b nt_w$w --workload notransition
done
b def_again; b nt_again --workload notransition
