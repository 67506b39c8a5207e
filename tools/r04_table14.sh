#!/bin/bash
# kernel times of the seed-table build for 14of22 (28-bit keys): three-level partition build vs the atomic counting sort
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04r; mkdir -p $out
for mode in partition atomic; do
  rm -rf /tmp/raw14
  (cd $R && SEGALIGN_AMD_TABLE_ATOMIC=$([ $mode = atomic ] && echo 1 || echo 0) rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw14 -o r -- python bench.py --seed 14of22 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/$mode.log 2>&1)
  python $R/tools/prof_summary.py /tmp/raw14 --out $out/table14_$mode.txt
  echo "== $mode"; grep -E "table_|scan_|nbr_|ctx_by" $out/table14_$mode.txt | head -14
done
