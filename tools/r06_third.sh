#!/bin/bash
# round 6, third GPU call: MAX_HITS iteration groups planned on the table-direct path -- parity first, then what a reference GPU's MAX_HITS costs now
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r06c; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_lookup_paths.py tests/test_gpu_h16.py tests/test_gpu_path_golden.py tests/test_gpu_rm_path_golden.py tests/test_gpu_edge_cases.py \
  tests/test_gpu_chain.py tests/test_gpu_join.py tests/test_gpu_filter_audit.py tests/test_gpu_rm_golden.py tests/test_gpu_rm_mask.py tests/test_gpu_rm_mask_grouped.py \
  tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_reader_golden.py tests/test_gpu_bench_contract.py -x -q > $out/tests_a.txt 2>&1; tail -6 $out/tests_a.txt
timeout 1500 python -m pytest tests/test_gpu_config_lumpy.py tests/test_gpu_config_human_block.py -x -q > $out/tests_b.txt 2>&1; tail -6 $out/tests_b.txt
for w in human lumpy; do
  for g in 8 15.78; do
    timeout 900 python bench.py --workload $w --max-hits-mem-gb $g --steps 3 --warmup 1 --no-dropin --no-cpu-baseline > $out/bench_line_${w}_maxhits_${g}.json 2> $out/bench_line_${w}_maxhits_${g}.err
  done
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-dropin --no-cpu-baseline > $out/bench_line_${w}.json 2> $out/bench_line_${w}.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c/bench_line_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", d["value"], "ms", d["ms_per_step"], "max_hits", d["max_hits"], "iters/step", d["reference_iterations_per_step"], "flags", d["path_flags"]["value"], "hsps", d["config"]["hsps_per_step"], "chk", d["config"]["hsp_checksum"])
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json", ".err")).read()[-1200:])
PY
