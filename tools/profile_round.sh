#!/bin/bash
# everything profiles/rN/ holds for one build, produced on the GPU box in one call:  PROFILE_TAG=r2_01 PROFILE_BUILD=<git hash> bash tools/profile_round.sh
R=$GRAFT_REPO_ROOT
T=${PROFILE_TAG:-r2}
B=${PROFILE_BUILD:-unknown}
O=$R/gpurun_out/p$T
mkdir -p $O
cd $R
python3 bench.py > $O/${T}_bench.json 2> $O/bench.err || echo "bench exit $?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_args.json 2>> $O/bench.err || echo "bench (driver args) exit $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/${T}_bench_under_rocprof.json 2> $O/kt.err || echo "rocprofv3 kernel-trace exit $?"
python3 $R/tools/prof_summary.py $O/kt > $O/${T}_kernel_summary.txt 2>&1 || true
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/nn_only.py 30 > $O/pf.log 2>&1 || echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/nn_only.py 30 > $O/pw.log 2>&1 || echo "pmc write exit $?"
python3 $R/tools/pmc_summary.py $O/pf $O/pw $O/${T}_pmc_hbm_traffic.json > $O/${T}_pmc_summary.txt 2>&1 || true
python3 - <<PY
import json
p = "$O/${T}_pmc_hbm_traffic.json"
try:
    d = json.load(open(p)); d["_build"] = "$B"; json.dump(d, open(p, "w"), indent=1)
except Exception as e:
    print("pmc json:", e)
PY
cd $R
python3 tools/nn_compare.py hall bunny grid128 big > $O/${T}_nn_compare_sparse_vs_dense.txt 2>&1
for row in 64 128; do
  ICP_NN_ROW=$row ICP_NN_PHASE_PASS=6 ICP_NN_PHASES=$O/ph.bin python3 tools/phase_run.py 9 > /dev/null 2>&1 && python3 tools/phase_report.py $O/ph.bin > $O/${T}_phase_log_resident_pass_rows_of_$row.txt
  [ $row = 64 ] && python3 tools/phase_sub.py $O/ph.bin 8 >> $O/${T}_phase_log_resident_pass_rows_of_$row.txt && python3 tools/cu_usage.py $O/ph.bin 8 > $O/${T}_cu_usage_rows_of_64.txt
  rm -f $O/ph.bin
  echo "== ICP_NN_ROW=$row" >> $O/${T}_reg_time.txt
  ICP_NN_ROW=$row python3 tools/reg_time.py 4000 >> $O/${T}_reg_time.txt 2>&1
  ICP_NN_ROW=$row python3 tools/reg_time.py 4000 plane >> $O/${T}_reg_time.txt 2>&1
done
python3 tools/r2_diag.py > $O/${T}_work_counters_one_registration.json 2> /dev/null
python3 bench.py --config s5 > $O/${T}_bench_s5_1gpu.json 2> $O/s5.err || echo "s5 exit $?"
# configs[1], Bunny.csv: the loop forms side by side (one box), what a pass costs early and late in a registration, the blocks of a late pass
python3 bench.py --config bunny > $O/${T}_bench_bunny.json 2> $O/bunny.err || echo "bunny exit $?"
{
  echo "# Bunny.csv 35 947^2, us per iteration of whole registrations (tools/bunny_time.py), same box:"
  VARIANTS="f a d i" bash tools/r2_share_ab.sh 2> /dev/null | grep -v "^\["
  if [ -f ab/libicp_r2_02.so ]; then echo "== previous build (r2_02: 16-wave blocks, 4096-entry hit list)"; ICP_LIB_PATH=$R/ab/libicp_r2_02.so python3 tools/bunny_time.py; fi
  echo "# what a pass costs (registrations of K fixed iterations): this build, then ICP_NN_WAVES128=16"
  python3 tools/bunny_marginal.py; ICP_NN_WAVES128=16 python3 tools/bunny_marginal.py
  echo "# other clouds of the size class (tools/grid_time.py): this build, then ICP_NN_WAVES128=16"
  python3 tools/grid_time.py; ICP_NN_WAVES128=16 python3 tools/grid_time.py
} > $O/${T}_bunny_shared_rows.txt 2>&1
PASSES="2 10 20" bash tools/r2_bunny_diag.sh > /dev/null 2>&1; grep -v "^phase [0-9]" gpurun_out/bunny_diag/phases.txt > $O/${T}_bunny_blocks_of_a_pass.txt
rm -rf $O/kt $O/pf $O/pw
ls -la $O
head -12 $O/${T}_kernel_summary.txt
head $O/${T}_pmc_summary.txt
