#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04y; mkdir -p $out
for t in noagg cur; do
  tree=$R/ab/$t; [ $t = cur ] && tree=$R
  for w in notransition ce11cb4; do
    rm -rf /tmp/raw_$t
    (cd $tree && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw_$t -o r -- python bench.py --workload $w --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --host-threads 1 --intervals-in-flight 1 > $out/${t}_$w.log 2>&1)
    python $R/tools/prof_summary.py /tmp/raw_$t --out $out/${t}_${w}_kernel_stats.txt
    echo "== $t $w"; grep -E "chain_|extend_filter_cls|probe_compact|extend_filter_packed" $out/${t}_${w}_kernel_stats.txt | head -7 | cut -c1-130
  done
done
