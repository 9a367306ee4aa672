set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p13
mkdir -p $O
cd $R
python bench.py > $O/13_bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/13_bench_under_rocprof.json 2> $O/kt.err || echo "rocprofv3 kernel-trace exit $?"
python3 $R/tools/prof_summary.py $O/kt > $O/13_kernel_summary.txt 2>&1 || true
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/13_kernel_stats_resident.csv || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/nn_only.py 30 > $O/pf.log 2>&1 || echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/nn_only.py 30 > $O/pw.log 2>&1 || echo "pmc write exit $?"
python3 $R/tools/pmc_summary.py $O/pf $O/pw $O/13_pmc_hbm_traffic_sparse.json > $O/13_pmc_summary.txt 2>&1 || true
cd $R
python tools/nn_compare.py hall bunny grid128 big > $O/13_nn_compare_sparse_vs_dense.txt 2>&1
ICP_NN_PHASE_PASS=5 ICP_NN_PHASES=$O/ph.bin python tools/phase_run.py 8 > /dev/null 2>&1 && python tools/phase_report.py $O/ph.bin > $O/13_phase_log_resident_pass.txt; rm -f $O/ph.bin
rm -rf $O/kt $O/pf $O/pw
ls -la $O
cat $O/13_kernel_summary.txt | head -12
cat $O/13_pmc_summary.txt | head
