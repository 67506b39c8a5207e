#!/bin/bash
# round 5: VALU issue rates of the class filter's instructions + the key-ordered join prototype (tools/micro/join_proto.hip)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r05b}; mkdir -p $out
timeout 300 $R/tools/micro/join_proto 12 73 3 1 > $out/join_check.txt 2>&1; cat $out/join_check.txt
timeout 900 $R/tools/micro/join_proto 24 73 3 0 > $out/join_q3.txt 2>&1; cat $out/join_q3.txt
