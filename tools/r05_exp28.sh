#!/bin/bash
# round 5 EXPERIMENT (timing only): the class filter reading its records at a 28-byte stride (ab_exp/libsegalign_hip.so = the same
# tree with extend.hip compiled with -DSA_EXP_REC28) against the real 32-byte records, interleaved on one box.  Build the variant here first:
#   mkdir -p ab_exp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSA_EXP_REC28 -c segalign_amd/csrc/extend.hip -o ab_exp/extend_exp.o && \
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ab_exp/libsegalign_hip.so $(ls segalign_amd/lib/obj/*.o | grep -v extend.o) ab_exp/extend_exp.o
# (ab_exp/ is git-ignored and travels with the gpurun snapshot.)  Result: profiles/r05/exp_rec28.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r05x; mkdir -p $out
rm -rf /tmp/exp; mkdir -p /tmp/exp; cp -r $R/segalign_amd $R/bench.py $R/oracle $R/profiles $R/tests /tmp/exp/ 2>/dev/null
cp $R/ab_exp/libsegalign_hip.so /tmp/exp/segalign_amd/lib/libsegalign_hip.so
for rep in 1 2 3; do
for e in 0 1; do
if [ $e = 1 ]; then cd /tmp/exp; else cd $R; fi
timeout 600 python bench.py --no-dropin --no-cpu-baseline --steps 10 --warmup 3 > $out/exp${e}_$rep.json 2> $out/exp${e}_$rep.err
python - <<PY
import json
d=json.loads(open("$out/exp${e}_$rep.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("exp28=$e rep $rep value %.4f ms %.2f filter_ss_us %.0f fwd %.4f" % (d["value"], d["ms_per_step"], r["single_stream"]["avg_launch_us"], r["per_hit"]["forwarded_frac"]))
PY
done; done
