#!/bin/bash
# round 5: SQ counters of the key-ordered class filter (join_filter_kernel) and of the kernels around it, one call in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05p}; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$TAG; mkdir -p $out
cd $R
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1 $*"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/raw_p$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/raw_p$i -o r -- python bench.py $ARGS > $out/bench_pmc$i.log 2>&1
  python tools/prof_summary.py /tmp/raw_p$i --out $out/pmc$i.txt
  rm -rf /tmp/raw_p$i
done
grep -A12 "join_filter" $out/pmc1.txt | head -14; grep -A12 "join_filter" $out/pmc2.txt | head -14
timeout 120 tools/micro/lds_rate > $out/lds_rate.txt 2>&1; cat $out/lds_rate.txt
