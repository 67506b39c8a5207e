#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04v; mkdir -p $out
$R/tools/micro/hit_shaped 3.2 5 17700000 > $out/timing_5.txt 2>&1; cat $out/timing_5.txt
$R/tools/micro/hit_shaped 34 78 3400000 > $out/timing_78.txt 2>&1; cat $out/timing_78.txt
for recs in 5 78; do
 for mode in 0 1 2 3; do
  rm -rf /tmp/raw_hs
  gb=3.2; n=17700000; [ $recs = 78 ] && gb=34 && n=3400000
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_HIT_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/raw_hs -o r -- $R/tools/micro/hit_shaped $gb $recs $n $mode > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/raw_hs --out $out/pmc_${recs}_$mode.txt
  echo "== recs $recs mode $mode (runs $n)"; grep -A5 "walk" $out/pmc_${recs}_$mode.txt | grep -E "RDREQ|MISS|HIT|READ_REQ"
 done
done
