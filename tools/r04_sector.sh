#!/bin/bash
# How many L2 -> fabric read requests does a short random piece cost, and how many bytes is a request?  (the gfx950
# doubling of FETCH_SIZE is calibrated on streams.)  piece_order reads N pieces of B bytes at random 32-byte-aligned offsets of a
# 34 GB buffer (order R), then sorted (S) and end to end (Q), each walk 1 warm-up + 3 timed launches, twice (two allocations).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04u; mkdir -p $out
for b in 32 64 128 160 256 2560; do
  n=20000000; [ $b = 2560 ] && n=3400000
  rm -rf /tmp/raw_sec
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum --output-format csv -d /tmp/raw_sec -o r -- $R/tools/micro/piece_order 34 $b $n > $out/piece_$b.log 2>&1
  python $R/tools/prof_summary.py /tmp/raw_sec --out $out/piece_$b.pmc.txt
  echo "== piece $b bytes, $n pieces per walk"; grep -A4 "walk" $out/piece_$b.pmc.txt | head -6
done
