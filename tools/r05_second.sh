#!/bin/bash
# round 5: new parity routes (rocThrust order chain, entropy switches, bench-size calls, rm default grouping, key-ordered split) + the grid tool's smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05h}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_thrust_order.py tests/test_gpu_entropy_options.py tests/test_gpu_join.py tests/test_gpu_rm_mask_grouped.py tests/test_gpu_multi_rank.py -x -q > $out/tests_new.txt 2>&1; tail -15 $out/tests_new.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "bench_default or one_full_chunk" > $out/tests_fullsize.txt 2>&1; tail -15 $out/tests_fullsize.txt
timeout 900 python tools/human_grid.py --grid 2 --target-mbp 40 --query-mbp 20 --check-bench --out $out/grid_smoke.json > /dev/null 2> $out/grid_smoke.err; tail -3 $out/grid_smoke.err
python - <<PY
import json
d=json.load(open("$out/grid_smoke.json"))
print({k:d[k] for k in ("grid_wall_s","compute_s","non_scaling_s","gbp_per_s","table_build_cold_s","table_build_warm_s","projection")}, d.get("bench_check"))
PY
