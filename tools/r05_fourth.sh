#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/${1:-r05j}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_block_switch.py tests/test_gpu_multi_device.py tests/test_gpu_edge_cases.py tests/test_gpu_lookup_paths.py tests/test_gpu_rm_golden.py tests/test_gpu_parity.py -q -x > $out/tests.txt 2>&1; tail -6 $out/tests.txt
SEGALIGN_AMD_DEBUG=1 timeout 1800 python tools/human_grid.py --grid ${GRID:-2} --out $out/grid_${GRID:-2}.json > /dev/null 2> $out/grid.err; grep -v "^pair\|granularity" $out/grid.err | tail -30
python - <<PY
import json
d=json.load(open("$out/grid_${GRID:-2}.json"))
print({k:d[k] for k in ("generate_s","grid_wall_s","compute_s","non_scaling_s","gbp_per_s","table_build_cold_s","table_build_warm_s","projection")})
for b in d["blocks"]: print(b)
PY
