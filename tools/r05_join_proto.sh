#!/bin/bash
# round 5: the key-ordered join prototype (tools/micro/join_proto.hip): selected variants at full size (q = mean query positions per key)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${1:-r05b}; mkdir -p $out
SEL=${2:-0xffffffff}
for q in ${3:-3}; do
timeout 900 $R/tools/micro/join_proto 24 ${4:-73} $q 0 $SEL > $out/join_q$q.txt 2>&1; cat $out/join_q$q.txt
done
