cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/thr_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/thr_tests.log
: > gpurun_out/thr_sweep.txt
for cfg in "512 1024" "768 1024" "768 512" "1024 512" "384 512" "640 512"; do
  set -- $cfg
  echo "== CTX_THREADS=$1 DEDUP_THREADS=$2" >> gpurun_out/thr_sweep.txt
  SEGALIGN_AMD_CTX_THREADS=$1 SEGALIGN_AMD_DEDUP_THREADS=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], json.dumps(j.get('kernels', {}))[:1500])
" >> gpurun_out/thr_sweep.txt
done
