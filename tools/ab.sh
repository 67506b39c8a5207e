#!/bin/bash
# ab.sh -- ONE same-box A/B driver (replaces the forty-odd one-off tools/r04_*.sh / tools/r05_*.sh scripts of rounds 4 and 5, which
# live on in git history: `git show f7273a5:tools/`).  Every box of the pool differs by a few per cent, so every comparison runs
# INTERLEAVED on one box: variant A rep 1, variant B rep 1, A rep 2, ...  Runs on the GPU box through gpurun; writes gpurun_out/<tag>/.
#
#   tools/ab.sh <tag> [--tests "<pytest args>"] [--reps N] [--steps K] [--warmup W] [--workloads "ce11cb4 lumpy ..."] [--roofline]
#               variant [variant ...]
#
# A variant is  name[:KEY=VALUE,KEY=VALUE...][:extra bench.py args]  -- environment first (SEGALIGN_AMD_<OPTION> = the engine's option
# table, GPU_MAX_HW_QUEUES, ...), bench arguments second.  Two special keys:
#     LIB=<path/to/libsegalign_hip.so>   run this variant from a copy of the tree (/tmp/ab_<name>) with that library in place of the
#                                        built one (an older commit's build kept under ab_exp/, which is git-ignored but travels)
#     TREE=<dir>                         run this variant from another checkout (ab/<commit>, built side by side)
# Examples
#     tools/ab.sh r06_l2 base l2r:SEGALIGN_AMD_L2_RIGHT_STATE=1                          an option against the default
#     tools/ab.sh r06_old --tests "tests/test_gpu_filter_audit.py -x -q" new old:LIB=ab_exp/libsegalign_hip_old.so
#     tools/ab.sh r06_cpc --workloads "ce11cb4 notransition" c40 c80::--chunks-per-call\ 80
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=${1:?tag}; shift
out=$R/gpurun_out/$tag; mkdir -p $out
reps=2; steps=10; warmup=3; workloads="ce11cb4"; tests=""; roof="--no-roofline"
while [ $# -gt 0 ]; do
  case "$1" in
    --tests) tests="$2"; shift 2;;
    --reps) reps=$2; shift 2;;
    --steps) steps=$2; shift 2;;
    --warmup) warmup=$2; shift 2;;
    --workloads) workloads="$2"; shift 2;;
    --roofline) roof=""; shift;;
    *) break;;
  esac
done
if [ -n "$tests" ]; then timeout 2400 python -m pytest $tests > $out/tests.txt 2>&1; tail -4 $out/tests.txt; fi
summary=$out/summary.txt; : > $summary
run() { # name dir envs args...
  local name=$1 dir=$2 envs=$3; shift 3
  local ev=(); IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && ev+=("$e"); done
  (cd $dir; env "${ev[@]}" timeout 1200 python bench.py --no-dropin --no-cpu-baseline $roof "$@" > $out/$name.json 2> $out/$name.err)
  python - "$out/$name.json" "$name" <<'PY' | tee -a $summary
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    ph, ss = r.get("per_hit") or {}, r.get("single_stream") or {}
    print("%-28s value %.4f  ms/step %8.2f  calls %3s  hsps %8d  chk %s  fwd %s  filter_ss_us %s  dom_share %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["config"]["calls_per_step"], d["config"]["hsps_per_step"], d["config"]["hsp_checksum"],
        ph.get("forwarded_frac"), ss.get("avg_launch_us"), r.get("dominant_share_of_gpu_time")))
except Exception as e:
    print("%-28s FAILED %s" % (sys.argv[2], e))
PY
}
for w in $workloads; do
  for rep in $(seq 1 $reps); do
    for v in "$@"; do
      name=${v%%:*}; rest=${v#*:}; [ "$rest" = "$v" ] && rest=""
      envs=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""
      dir=$R; keep=""
      IFS=',' read -ra kv <<< "$envs"
      for e in "${kv[@]}"; do
        case "$e" in
          LIB=*) dir=/tmp/ab_$name
                 if [ ! -d $dir ]; then mkdir -p $dir; cp -r $R/segalign_amd $R/bench.py $R/oracle $R/profiles $R/tests $dir/ 2> /dev/null; cp $R/${e#LIB=} $dir/segalign_amd/lib/libsegalign_hip.so; fi;;
          TREE=*) dir=$R/${e#TREE=};;
          *) keep="$keep,$e";;
        esac
      done
      run ${w}_${name}_$rep $dir "${keep#,}" --workload $w --steps $steps --warmup $warmup $extra
    done
  done
done
echo "== $summary"
