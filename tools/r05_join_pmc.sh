#!/bin/bash
# round 5: where do the cycles of the join prototype go?  (SQ counters, two passes; variants 0, 5, 8, 14 of tools/micro/join_proto.hip)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05c; mkdir -p $out
SEL=${1:-0x4121}
rm -rf /tmp/raw_a /tmp/raw_b
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/raw_a -o r -- $R/tools/micro/join_proto 24 73 3 0 $SEL > $out/run_a.txt 2>&1
python $R/tools/prof_summary.py /tmp/raw_a --out $out/pmc_a.txt
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/raw_b -o r -- $R/tools/micro/join_proto 24 73 3 0 $SEL > $out/run_b.txt 2>&1
python $R/tools/prof_summary.py /tmp/raw_b --out $out/pmc_b.txt
cat $out/run_a.txt; grep -A12 "join_filter" $out/pmc_a.txt; grep -A12 "join_filter" $out/pmc_b.txt
