#!/bin/bash
# End-of-round check on the GPU box (tools/final_check.sh [tag]): the whole GPU suite with durations + smoke on the tree as it is, the driver's bench
# command, the other workloads' lines with a longer warm-up.  Writes gpurun_out/<tag>/{gputests_final,smoke_final}.txt and bench_line_*.json.
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r06}; out=gpurun_out/$tag; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $out/gputests_final.txt 2>&1; tail -40 $out/gputests_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_final.txt 2>&1; tail -2 $out/smoke_final.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_line_driver_args.json 2> $out/bench_line_driver_args.err
for w in notransition rm lumpy human; do
  timeout 900 python bench.py --workload $w --steps 6 --warmup 3 > $out/bench_line_${w}_warm.json 2> /dev/null
done
python - $tag <<'PY'
import json, glob, sys
tag = sys.argv[1]
for f in ["bench_line_driver_args"] + ["bench_line_%s_warm" % w for w in ("notransition", "rm", "lumpy", "human")]:
    try:
        d = json.loads(open("gpurun_out/%s/%s.json" % (tag, f)).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "drained", d["config"]["ms_per_pass_drained"], "frac", r["frac"], "ss", r.get("single_stream_frac"), "busy", r["busy_share_of_timed_region"], "upl", (d.get("query_upload_inclusive") or {}).get("value"), "check", r["profile_check"]["ok"])
    except Exception as e:
        print(f, "failed", e)
PY
