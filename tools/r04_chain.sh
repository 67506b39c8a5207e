#!/bin/bash
# chain grouping: buckets sized on the device by candidates (target per bucket) against forced bucket counts (round 4's by-hits rule
# gave 16384 for a --notransition call, 32768 for a default one); LDS entries per sort + link workgroup (occupancy).  Same box,
# interleaved whole-bench runs, then single-stream kernel durations under rocprofv3.
out=$PWD/gpurun_out/r04x; mkdir -p $out
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --steps 6 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    d=json.load(open("$out/$name.json")); r=d["roofline"]; k=r["kernels"]
    print("$name", d["value"], d["ms_per_step"], "chain_group us (in flight)", k["chain_group"]["avg_us"], "exact_chain", k["extend_exact_chain"]["avg_us"], d["config"]["hsp_checksum"])
except Exception as e: print("$name FAILED", e)
PY
}
for rep in 1 2; do
  SEGALIGN_AMD_CHAIN_BUCKETS=16384 SEGALIGN_AMD_CHAIN_SORT_BLOCKS=2048 SEGALIGN_AMD_CHAIN_GROUP_MAX=4096 b nt_old_$rep --workload notransition
  for g in 1024 2048; do SEGALIGN_AMD_CHAIN_GROUP_MAX=$g b nt_g${g}_$rep --workload notransition; done
  SEGALIGN_AMD_CHAIN_BUCKETS=32768 SEGALIGN_AMD_CHAIN_GROUP_MAX=4096 b def_old_$rep
  SEGALIGN_AMD_CHAIN_GROUP_MAX=1024 b def_g1024_$rep
done
cd /tmp && export TMPDIR=/tmp
prof() { v=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-dropin --steps 3 --warmup 1 --host-threads 1 --intervals-in-flight 1 > $out/prof_$v.json 2> $out/prof_$v.err
  f=$(find $out/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sa::chain" in r["Name"] or "exact_chain" in r["Name"]: print(" ", r["Name"][:50], r["Calls"], r["AverageNs"])
PY
  rm -rf $out/prof_$v
}
SEGALIGN_AMD_CHAIN_BUCKETS=16384 SEGALIGN_AMD_CHAIN_SORT_BLOCKS=2048 SEGALIGN_AMD_CHAIN_GROUP_MAX=4096 prof nt_old --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=4096 prof nt_g4096 --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=2048 prof nt_g2048 --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=1024 prof nt_g1024 --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=1024 SEGALIGN_AMD_CHAIN_SORT_THREADS=128 prof nt_g1024_t128 --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=512 SEGALIGN_AMD_CHAIN_BUCKET_TARGET=16 prof nt_g512_t16 --workload notransition
SEGALIGN_AMD_CHAIN_BUCKETS=32768 SEGALIGN_AMD_CHAIN_GROUP_MAX=4096 prof def_old
SEGALIGN_AMD_CHAIN_GROUP_MAX=1024 prof def_g1024
