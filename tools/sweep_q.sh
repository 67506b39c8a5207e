#!/bin/bash
# A/B on one box, interleaved: HW queues x engine slots x host threads (calls in flight = threads x 2 intervals).
# "default" = nothing set: the library's constructor asks for 8 queues, 4 slots, bench issues 3 calls per interval
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  v=$(timeout 200 python bench.py --no-dropin --no-cpu-baseline --no-roofline --steps 6 < /dev/null 2>/dev/null | cut -c60-80)
  echo "default -> $v"
  for cfg in "4 2 2" "8 4 3" "12 4 3"; do
    set -- $cfg
    v=$(GPU_MAX_HW_QUEUES=$1 SEGALIGN_AMD_SLOTS=$2 timeout 200 python bench.py --no-dropin --no-cpu-baseline --no-roofline --host-threads $3 --steps 6 < /dev/null 2>/dev/null | cut -c60-80)
    echo "hwq=$1 slots=$2 threads=$3 -> $v"
  done
done
