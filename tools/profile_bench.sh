#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration ON THE GPU BOX (run through gpurun).
#   tools/profile_bench.sh <tag> [bench args for the --kernel-trace --stats pass...]
# Writes small text summaries to gpurun_out/<tag>/ (raw CSVs are deleted: gpurun copies back <= 64 MiB).
# Counter passes are SEPARATE runs from the --kernel-trace --stats pass (MI355X_MICROARCH.md, rocprofv3 PMC slots:
# SQ 8 / TCC 4 / GRBM 2 per pass; FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> one pass each).
set -u
TAG=$1; shift
PMC_ARGS=${PMC_ARGS:---steps 1 --warmup 0 --no-cpu-baseline}
WORKLOAD=${WORKLOAD:-ce11cb4}; TARGET_MBP=${TARGET_MBP:-100.0}; CHUNK=${CHUNK:-250000}   # key of the profiled workload shape
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd $R
echo "stats pass: python bench.py $*" > $OUT/commands.txt
# bench.py quotes traffic.json only for this shape; single_stream: the stats pass ran with ONE call in flight (SINGLE_STREAM=1 and
# "--host-threads 1 --intervals-in-flight 1" among the bench args), so its kernel averages are overlap-free
python -c "import json,sys; json.dump({'workload': sys.argv[1], 'target_mbp': float(sys.argv[2]), 'chunk': int(sys.argv[3]), 'single_stream': sys.argv[5] == '1'}, open(sys.argv[4], 'w'))" $WORKLOAD $TARGET_MBP $CHUNK $OUT/workload.json ${SINGLE_STREAM:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r -- python bench.py "$@" > $OUT/bench_under_stats.log 2>&1
python tools/prof_summary.py $RAW/stats --out $OUT/kernel_stats.txt
grep "^{\"metric\"" $OUT/bench_under_stats.log | tail -1 > $OUT/bench_line_under_stats.json
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  echo "pmc$i: --pmc $set -- python bench.py $PMC_ARGS" >> $OUT/commands.txt
  rocprofv3 --pmc $set --output-format csv -d $RAW/pmc$i -o r -- python bench.py $PMC_ARGS > $OUT/bench_under_pmc$i.log 2>&1
  python tools/prof_summary.py $RAW/pmc$i --out $OUT/pmc$i.txt
done
python tools/traffic_json.py $OUT/pmc3.txt $OUT/pmc4.txt > $OUT/traffic.json
rm -rf $RAW
ls -la $OUT
