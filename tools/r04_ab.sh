#!/bin/bash
# round-4 A/B on ONE box: call sizing and slots for the sparse-hit workload, default workload before/after, lumpy
out=gpurun_out/r04c; mkdir -p $out
b() { name=$1; shift; python bench.py --no-roofline --no-cpu-baseline --steps 6 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"]["calls_per_step"], d["config"]["hsps_per_step"], d["config"]["hsp_checksum"])
except Exception as e: print("$name FAILED", e)
PY
}
b default
SEGALIGN_AMD_CALL_HITS=0 b notrans_20chunks --workload notransition
b notrans_sized --workload notransition
b notrans_sized_8inflight --workload notransition --host-threads 4
SEGALIGN_AMD_CALL_HITS=$((128<<20)) b notrans_128M --workload notransition
SEGALIGN_AMD_CALL_HITS=$((512<<20)) b notrans_512M --workload notransition
b default_8inflight --host-threads 4
b lumpy --workload lumpy --steps 2 --warmup 1
