#!/bin/bash
# round 6, second GPU call: parity of the right-walk hand-over (option l2_right_state), then same-box A/Bs: the hand-over, stream priorities
# (option filter_prio), hardware queues; the run-alignment micro (verdict item 6).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r06b; mkdir -p $out
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o run_align run_align.hip) 2> $out/run_align.build.err
timeout 600 tools/micro/run_align > $out/run_align.txt 2>&1; cat $out/run_align.txt
tools/ab.sh r06b --tests "tests/test_gpu_filter_audit.py tests/test_gpu_parity.py tests/test_gpu_lookup_paths.py tests/test_gpu_join.py tests/test_gpu_block_edges.py tests/test_gpu_random.py tests/test_gpu_edge_cases.py tests/test_gpu_rm_golden.py tests/test_gpu_find_hsps_golden.py tests/test_gpu_chain.py -x -q" \
  --reps 2 --roofline --workloads "ce11cb4 lumpy notransition" \
  base l2r0:SEGALIGN_AMD_L2_RIGHT_STATE=0 prio:SEGALIGN_AMD_FILTER_PRIO=1,GPU_MAX_HW_QUEUES=16 q16:GPU_MAX_HW_QUEUES=16
