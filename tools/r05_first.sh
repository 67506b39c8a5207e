#!/bin/bash
# round 5, first GPU call of the session: key-ordered calls -- parity tests, then the default pass with and without them on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_join.py -x -q > $out/test_join.txt 2>&1; tail -5 $out/test_join.txt
for rep in 1 2; do
for ko in 1 0; do
SEGALIGN_AMD_KEY_ORDER=$ko timeout 600 python bench.py --steps 10 --warmup 3 --no-dropin > $out/bench_ko${ko}_$rep.json 2> $out/bench_ko${ko}_$rep.err
python - <<PY
import json
try:
    d=json.loads(open("$out/bench_ko${ko}_$rep.json").read().strip().splitlines()[-1])
    print("ko=$ko rep=$rep value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"].get("frac"), d["roofline"].get("kernel"))
except Exception as e:
    print("ko=$ko failed", e); print(open("$out/bench_ko${ko}_$rep.err").read()[-2000:])
PY
done; done
