#!/bin/bash
# single-stream per-scope kernel averages of the default workload (one call in flight): quick look between two edits
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python bench.py --no-dropin --no-cpu-baseline --host-threads 1 --intervals-in-flight 1 --steps 2 --warmup 1 "$@" < /dev/null 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('solo', l['value'], l['ms_per_step'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
