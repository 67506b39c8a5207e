#!/bin/bash
# configs[2] rehearsal on one GPU: ONE 200 Mbp x 80 Mbp block pair, its calls dealt to 1 and to 2 ranks that share HIP device 0
# (gloo for the barrier / reductions): same checksum, HSP count and bases; the timing says nothing about scaling
out=gpurun_out/r04s; mkdir -p $out
common="--workload human --target-mbp 200 --chunks-per-call 10 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --backend gloo --share-gpu"
SEGALIGN_AMD_ARENA_GB=80 python bench.py --gpus 1 $common > $out/human200_w1.json 2> $out/w1.err
SEGALIGN_AMD_ARENA_GB=80 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 $common > $out/human200_w2.json 2> $out/w2.err
python - <<PY
import json
a=json.load(open("$out/human200_w1.json")); b=[json.loads(l) for l in open("$out/human200_w2.json") if l.startswith("{")][-1]
for d in (a,b):
    c=d["config"]; print(d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], c["calls_per_step"], c["hsps_per_step"], c["query_bases_per_step"], c["hsp_checksum"], c["partition_imbalance"], c["partition_cost_ms"])
print("SAME" if (a["config"]["hsp_checksum"], a["config"]["hsps_per_step"]) == (b["config"]["hsp_checksum"], b["config"]["hsps_per_step"]) else "DIFFERENT")
PY
