#!/bin/bash
# latency-hiding variants of the context filter (SEGALIGN_AMD_CTX_PIPE) x workgroup size
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 env SEGALIGN_AMD_CTX_PIPE=1 SEGALIGN_AMD_CTX_THREADS=1024 python -m pytest tests/test_gpu_lookup_paths.py -m gpu -x -q > gpurun_out/pipe_tests.log 2>&1; echo "rc=$?" >> gpurun_out/pipe_tests.log
bash tools/sweep_bench.sh "SEGALIGN_AMD_CTX_PIPE=2" "SEGALIGN_AMD_CTX_PIPE=1 SEGALIGN_AMD_CTX_THREADS=1024" "SEGALIGN_AMD_CTX_PIPE=1 SEGALIGN_AMD_CTX_THREADS=512" "SEGALIGN_AMD_CTX_PIPE=1 SEGALIGN_AMD_CTX_THREADS=256" > gpurun_out/sweep_pipe.txt 2>&1
