#!/bin/bash
# per-kernel comparison of two trees on one box: rocprofv3 kernel stats of a single-stream sparse-hit pass
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04g; mkdir -p $out
for t in pre cur; do
  tree=$R/ab/$t; [ $t = cur ] && tree=$R
  for w in notransition ce11cb4; do
    rm -rf /tmp/raw_$t
    (cd $tree && SEGALIGN_AMD_CALL_HITS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw_$t -o r -- python bench.py --workload $w --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --host-threads 1 --intervals-in-flight 1 > $out/${t}_$w.log 2>&1)
    python $R/tools/prof_summary.py /tmp/raw_$t --out $out/${t}_${w}_kernel_stats.txt
  done
done
ls $out
