#!/bin/bash
# profiles/rN evidence for every BASELINE config, produced on the GPU box in one call:
#   PROFILE_TAG=r3_01 PROFILE_BUILD=<git hash> bash tools/profile_configs.sh [configs...]
# per config: the bench line (plain run), the line printed under rocprofv3 --kernel-trace --stats with the per-kernel summary
# of that run, and two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) -> HBM bytes per launch.
R=$GRAFT_REPO_ROOT
T=${PROFILE_TAG:-r3}
B=${PROFILE_BUILD:-unknown}
O=$R/gpurun_out/p$T
mkdir -p $O
CFGS=${@:-hall hall_plane bunny s5 cpu_f64}
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  echo "== $c"
  (cd $R && python3 bench.py --config $c > $O/${T}_bench_$c.json 2> $O/bench_$c.err) || echo "bench $c exit $?"
  # the measuring process itself behind the profiler (bench.py's supervisor would only add a hop)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- python3 $R/bench.py --leg main --config $c --no-cpu-baseline > $O/${T}_bench_${c}_under_rocprof.json 2> $O/kt_$c.err || echo "rocprofv3 kernel-trace $c exit $?"
  python3 $R/tools/prof_summary.py $O/kt_$c > $O/${T}_kernel_summary_$c.txt 2>&1 || true
  cp $(find $O/kt_$c -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_$c.csv 2> /dev/null || true
  # HBM traffic of the config's kernels: short runs (every launch is serialised under --pmc)
  case $c in
    hall|hall_plane|cpu_f64) PARGS="--steps 60 --warmup 0" ;;
    bunny) PARGS="--steps 44 --warmup 0" ;;
    s5) PARGS="--steps 10 --warmup 0" ;;
  esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf_$c -- python3 $R/bench.py --leg main --config $c --no-cpu-baseline $PARGS > $O/pf_$c.log 2>&1 || echo "pmc fetch $c exit $?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw_$c -- python3 $R/bench.py --leg main --config $c --no-cpu-baseline $PARGS > $O/pw_$c.log 2>&1 || echo "pmc write $c exit $?"
  python3 $R/tools/pmc_summary.py $O/pf_$c $O/pw_$c $O/${T}_pmc_hbm_traffic_$c.json > $O/${T}_pmc_summary_$c.txt 2>&1 || true
  python3 - <<PY
import json
p = "$O/${T}_pmc_hbm_traffic_$c.json"
try:
    d = json.load(open(p)); d["_build"] = "$B"; d["_what"] = "average over the launches of bench.py --leg main --config $c $PARGS"
    json.dump(d, open(p, "w"), indent=1)
except Exception as e:
    print("pmc json:", e)
PY
  rm -rf $O/kt_$c $O/pf_$c $O/pw_$c
  head -6 $O/${T}_kernel_summary_$c.txt
  grep nn_match $O/${T}_pmc_summary_$c.txt | head -4
done
ls -la $O
