cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl2
timeout 900 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_config_human_block.py tests/test_gpu_config_plumbing.py -m gpu -x -q > gpurun_out/wl2/tests.log 2>&1; echo "rc=$?" >> gpurun_out/wl2/tests.log
python bench.py --no-cpu-baseline > gpurun_out/wl2/bench_default.json 2> gpurun_out/wl2/bench_default.err
for w in human notransition rm plumbing; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/wl2/bench_$w.json 2> gpurun_out/wl2/bench_$w.err
done
