#!/bin/bash
# tools/sweep_bench.sh "<ENV1=..> <ENV2=..>" ... -- one short bench run per environment set (GPU box); prints value + the main kernels
# extra bench arguments via BENCH_ARGS
cd ${GRAFT_REPO_ROOT:-/root/repo}
for envset in "$@"; do
  out=$(env $envset python bench.py --no-cpu-baseline --steps 2 --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1)
  python - "$envset" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]; k = r["kernels"]
top = sorted(k.items(), key=lambda kv: -kv[1]["ms_total"])[:7]
print("%-60s %.4f Gbp/s  %7.1f ms/step  single %s us | %s" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"],
      (r.get("single_stream") or {}).get("avg_launch_us"), " ".join("%s=%.0f" % (n, v["avg_us"]) for n, v in top)))
PY
done
