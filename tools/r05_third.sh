#!/bin/bash
# round 5: remaining new tests + the 2 x 2 grid of 500 Mbp blocks
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05i}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_thrust_order.py tests/test_gpu_entropy_options.py tests/test_gpu_join.py tests/test_gpu_rm_mask_grouped.py tests/test_gpu_multi_rank.py -q > $out/tests_new.txt 2>&1; tail -8 $out/tests_new.txt
timeout 1800 python tools/human_grid.py --grid ${GRID:-2} --check-bench --out $out/grid_${GRID:-2}.json > /dev/null 2> $out/grid.err; tail -5 $out/grid.err
python - <<PY
import json
d=json.load(open("$out/grid_${GRID:-2}.json"))
print({k:d[k] for k in ("generate_s","grid_wall_s","compute_s","non_scaling_s","gbp_per_s","table_build_cold_s","table_build_warm_s","projection")}, d.get("bench_check"))
for b in d["blocks"]: print(b)
for p in d["pairs"]: print(p)
PY
