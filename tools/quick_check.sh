#!/bin/bash
# quick GPU confidence run after a kernel change: lookup-path + parity + golden extension tests, then two bench runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_parity.py tests/test_gpu_find_hsps_golden.py -m gpu -x -q > gpurun_out/quick_tests.log 2>&1; echo "rc=$?" >> gpurun_out/quick_tests.log
bash tools/sweep_bench.sh "" "SEGALIGN_AMD_CTX_THREADS=512" > gpurun_out/quick_bench.txt 2>&1
