cd $GRAFT_REPO_ROOT
{
for ht in 1 2 3 4; do echo "## host-threads $ht"; BENCH_ARGS="--host-threads $ht" bash tools/sweep_bench.sh "SEGALIGN_AMD_SLOTS=4"; done
bash tools/sweep_bench.sh "SEGALIGN_AMD_L2_BLOCKS=128" "SEGALIGN_AMD_L2_BLOCKS=512" "SEGALIGN_AMD_L2_BLOCKS=1024" "SEGALIGN_AMD_CHAIN_SORT_THREADS=512" "SEGALIGN_AMD_CHUNKS_PER_CALL=2" "SEGALIGN_AMD_CHUNKS_PER_CALL=3"
} > gpurun_out/sweep_knobs.txt 2>&1
