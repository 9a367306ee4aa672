set -e
R=$GRAFT_REPO_ROOT
T=${PROFILE_TAG:-13}
O=$R/gpurun_out/p$T
mkdir -p $O
cd $R
python bench.py > $O/${T}_bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/${T}_bench_under_rocprof.json 2> $O/kt.err || echo "rocprofv3 kernel-trace exit $?"
python3 $R/tools/prof_summary.py $O/kt > $O/${T}_kernel_summary.txt 2>&1 || true
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_resident.csv || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/nn_only.py 30 > $O/pf.log 2>&1 || echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/nn_only.py 30 > $O/pw.log 2>&1 || echo "pmc write exit $?"
python3 $R/tools/pmc_summary.py $O/pf $O/pw $O/${T}_pmc_hbm_traffic_sparse.json > $O/${T}_pmc_summary.txt 2>&1 || true
cd $R
python tools/nn_compare.py hall bunny grid128 big > $O/${T}_nn_compare_sparse_vs_dense.txt 2>&1
ICP_NN_PHASE_PASS=5 ICP_NN_PHASES=$O/ph.bin python tools/phase_run.py 8 > /dev/null 2>&1 && python tools/phase_report.py $O/ph.bin > $O/${T}_phase_log_resident_pass.txt; rm -f $O/ph.bin
rm -rf $O/kt $O/pf $O/pw
ls -la $O
cat $O/${T}_kernel_summary.txt | head -12
cat $O/${T}_pmc_summary.txt | head
