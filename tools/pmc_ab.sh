#!/bin/bash
# counters of the extension filter variants on the bench workload (GPU box): tools/pmc_ab.sh <tag> [env assignments...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_raw
  env "$@" rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_raw -o r -- python tools/filter_ab.py 100 > $OUT/log$i.txt 2>&1
  python tools/prof_summary.py /tmp/pmc_raw --out $OUT/pmc$i.txt
done
grep -h -A9 "extend_filter" $OUT/pmc*.txt | grep -v "^--" > $OUT/filter_counters.txt
cat $OUT/filter_counters.txt
