#!/bin/bash
# chunks per multi-chunk call: parity of the multi-chunk paths first, then the bench at 4 / 8 / 12 / 16 chunks per call
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/chunks_tests.log 2>&1; echo "rc=$?" >> gpurun_out/chunks_tests.log
bash tools/sweep_bench.sh "SEGALIGN_AMD_CHUNKS_PER_CALL=4" "SEGALIGN_AMD_CHUNKS_PER_CALL=8" "SEGALIGN_AMD_CHUNKS_PER_CALL=10" "SEGALIGN_AMD_CHUNKS_PER_CALL=16" "SEGALIGN_AMD_CHUNKS_PER_CALL=8 SEGALIGN_AMD_SLOTS=3" > gpurun_out/sweep_chunks.txt 2>&1
BENCH_ARGS="--host-threads 3" bash tools/sweep_bench.sh "SEGALIGN_AMD_CHUNKS_PER_CALL=8 SEGALIGN_AMD_SLOTS=3" >> gpurun_out/sweep_chunks.txt 2>&1
