#!/bin/bash
# counters of chain_sort_link_kernel on one single-stream pass of the sparse-hit workload
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04x_pmc; mkdir -p $OUT
ARGS="${WL:---workload notransition} --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --host-threads 1 --intervals-in-flight 1"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAVES_EQ_64"; do
  i=$((i+1)); rm -rf /tmp/pmc_raw
  ( cd $R && timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_raw -o r -- python bench.py $ARGS > $OUT/run$i.log 2>&1 )
  python $R/tools/prof_summary.py /tmp/pmc_raw --out $OUT/pmc$i.txt
  grep -A10 "chain_sort_link" $OUT/pmc$i.txt | head -11
done
rm -rf /tmp/pmc_raw
