cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl3
timeout 900 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_config_human_block.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/wl3/tests.log 2>&1; echo "rc=$?" >> gpurun_out/wl3/tests.log
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/wl3/bench_default.json 2> gpurun_out/wl3/bench_default.err
for w in human plumbing; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/wl3/bench_$w.json 2> gpurun_out/wl3/bench_$w.err
done
