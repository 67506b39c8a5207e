#!/bin/bash
# Counter passes for the context filter on ONE interval of the default workload (its calls of the engine's default size, one call in flight):
# each --pmc pass serialises the kernels, so the command is kept short.  Usage (GPU box): [KERNEL=name] [OUTTAG=dir] bash tools/pmc_ctx.sh
KERNEL=${KERNEL:-extend_filter_cls_kernel}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUTTAG:-pmc_ctx}
mkdir -p $OUT
rm -rf /tmp/pmc_raw
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_raw -o r -- python bench.py --one-interval > $OUT/run0.log 2>&1 )
python $R/tools/prof_summary.py /tmp/pmc_raw --out $OUT/kernel_stats.txt
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TA_BUSY_AVG TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_raw
  ( cd $R && timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_raw -o r -- python bench.py --one-interval > $OUT/run$i.log 2>&1 )
  python $R/tools/prof_summary.py /tmp/pmc_raw --out $OUT/pmc$i.txt
  grep -A10 "$KERNEL" $OUT/pmc$i.txt | head -11 > $OUT/ctx$i.txt
done
rm -rf /tmp/pmc_raw
