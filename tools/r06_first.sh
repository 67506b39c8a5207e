#!/bin/bash
# round 6, first GPU call: the class-walk micro (verdict r05 item 1a), the new parity tests (MAX_HITS of the reference GPUs at workload
# density, the oracle's own table at 100 Mbp, 8 engine devices x 6 slots), the driver's bench line, the owned host end to end.
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r06; mkdir -p $out
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o class_walk class_walk.hip) 2> $out/class_walk.build.err
for rep in 1 2; do timeout 300 tools/micro/class_walk; done > $out/class_walk.txt 2>&1
tail -8 $out/class_walk.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi_device.py tests/test_gpu_bench_contract.py -x -q > $out/tests_a.txt 2>&1; tail -5 $out/tests_a.txt
timeout 1200 python -m pytest tests/test_gpu_config_lumpy.py tests/test_gpu_config_human_block.py -x -q --durations=8 > $out/tests_b.txt 2>&1; tail -14 $out/tests_b.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_line_driver_args.json 2> $out/bench_line_driver_args.err; tail -c 600 $out/bench_line_driver_args.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_line_driver_args.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "upload-inclusive", d.get("query_upload_inclusive"), "frac", r["frac"], "ss", r["single_stream_frac"], "dom", r["dominant_share_of_gpu_time"], "check", r["profile_check"]["ok"])
PY
timeout 1200 python tools/run_host_fullsize.py $out/host_e2e.txt > $out/host_e2e.log 2>&1; tail -22 $out/host_e2e.log
timeout 900 python bench.py --workload human --max-hits-mem-gb 8 --steps 2 --warmup 1 --no-dropin --no-cpu-baseline > $out/bench_line_human_maxhits_m60.json 2> $out/bench_line_human_maxhits_m60.err
timeout 900 python bench.py --workload human --steps 2 --warmup 1 --no-dropin --no-cpu-baseline > $out/bench_line_human.json 2> $out/bench_line_human.err
python - <<'PY'
import json
for n in ("bench_line_human_maxhits_m60", "bench_line_human"):
    try:
        d=json.loads(open("gpurun_out/r06/%s.json" % n).read().strip().splitlines()[-1])
        print(n, "value", d["value"], "ms", d["ms_per_step"], "max_hits", d["max_hits"], "iters/step", d["reference_iterations_per_step"], "flags", d["path_flags"], "hsps", d["config"]["hsps_per_step"], "chk", d["config"]["hsp_checksum"])
    except Exception as e:
        print(n, "failed", e); print(open("gpurun_out/r06/%s.err" % n).read()[-1500:])
PY
