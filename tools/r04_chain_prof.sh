#!/bin/bash
# single-stream durations of the chain kernels under rocprofv3, per option set (second half of r04_chain.sh)
out=$PWD/gpurun_out/r04x; mkdir -p $out
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
prof nt_new --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=512 prof nt_g512 --workload notransition
SEGALIGN_AMD_CHAIN_GROUP_MAX=2048 prof nt_g2048 --workload notransition
SEGALIGN_AMD_CHAIN_SORT_BLOCKS=16384 prof nt_b16384 --workload notransition
SEGALIGN_AMD_CHAIN_BUCKET_TARGET=64 prof nt_t64 --workload notransition
prof def_new
SEGALIGN_AMD_CHAIN_GROUP_MAX=512 prof def_g512
prof lumpy_new --workload lumpy
