#!/bin/bash
# Memory-side counter passes (address translation, L1/TA/TD busy, latencies) for one kernel on ONE interval of the default workload.
# Usage (GPU box): [KERNEL=name] [OUTTAG=dir] bash tools/pmc_mem.sh
KERNEL=${KERNEL:-extend_filter_cls_kernel}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUTTAG:-pmc_mem}
mkdir -p $OUT
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_raw
  ( cd $R && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_raw -o r -- python bench.py --one-interval > $OUT/run$i.log 2>&1 < /dev/null )
  python $R/tools/prof_summary.py /tmp/pmc_raw --out $OUT/pmc$i.txt
  grep -A10 "$KERNEL" $OUT/pmc$i.txt | head -11 > $OUT/k$i.txt
  cat $OUT/k$i.txt
done
rm -rf /tmp/pmc_raw $OUT/run*.log
