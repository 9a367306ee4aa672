#!/bin/bash
# the rest of profiles/rN for one build, in one call:  PROFILE_TAG=r3_02 bash tools/profile_extras.sh
#   sweep CSVs of the reference (NUM_POINTS,TIME), multi-rank rehearsals on the one GPU (shared-memory route, RCCL leg), the
#   driver's own command line, hall phase logs / registration times (8- and 16-wave rows), probes, dense-kernel sweep, short soak
R=$GRAFT_REPO_ROOT; T=${PROFILE_TAG:-r3}; O=$R/gpurun_out/x$T; mkdir -p $O/sweeps
cd $R
B=fast-point-cloud-registration-with-gpus_amd/bin
(cd $O/sweeps && $R/$B/ICP_time_complexity > /dev/null 2>&1; $R/$B/ICP_time_complexity --plane > /dev/null 2>&1; $R/$B/ICP_time_complexity --matching > /dev/null 2>&1; tail -1 *.csv)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_args.json 2> $O/driver.err || echo "driver-args bench exit $?"
ICP_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 2 > $O/${T}_bench_hall_2ranks_on_1gpu_rehearsal.json 2> $O/spawn2.err; echo "rc=$?" >> $O/spawn2.err
ICP_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline > $O/${T}_bench_hall_rccl_leg_1rank.json 2> $O/rccl1.err; echo "rc=$?" >> $O/rccl1.err
ICP_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --config s5 --no-cpu-baseline > $O/${T}_bench_s5_rccl_leg_1rank.json 2> $O/rccl_s5.err; echo "rc=$?" >> $O/rccl_s5.err
ICP_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --config s5 --gpus 2 --no-cpu-baseline > $O/${T}_bench_s5_2ranks_on_1gpu_rehearsal.json 2> $O/s5_2r.err; echo "rc=$?" >> $O/s5_2r.err
tail -2 $O/*.err
{
  for w in 8 16; do
    echo "== hall, ICP_NN_WAVES=$w: registrations back to back (tools/reg_time.py), point-to-point and point-to-plane"
    ICP_NN_WAVES=$w python3 tools/reg_time.py 4000; ICP_NN_WAVES=$w python3 tools/reg_time.py 4000 plane
  done
} > $O/${T}_reg_time.txt 2>&1
for w in 8 16; do
  ICP_NN_WAVES=$w ICP_NN_PHASES=$O/ph.bin:6 python3 tools/phase_run.py 9 > /dev/null 2>&1 && python3 tools/phase_report.py $O/ph.bin $w > $O/${T}_phase_log_resident_pass_${w}_waves.txt
  rm -f $O/ph.bin
done
python3 tools/bunny_first.py > $O/${T}_bunny_first_registration.txt 2>&1
for w in hall bunny bunny_res grid128; do python3 tools/setup_time.py $w; done > $O/${T}_setup_time.txt 2>&1
[ -x bin/mfma_filter_probe ] && bin/mfma_filter_probe > $O/${T}_mfma_filter_probe.txt 2>&1
python3 tools/s5_time.py 0 8 30 > $O/${T}_s5_share_rank0_of_8.txt 2>&1
python3 tools/work_counters.py > $O/${T}_work_counters_one_registration.json 2> /dev/null
python3 tools/nn_compare.py hall bunny grid128 big > $O/${T}_nn_compare_sparse_vs_dense.txt 2>&1
ICP_NN_SPARSE=0 ICP_NN_CULL=0 ICP_NN_PHASES=$O/ph.bin python3 tools/dense_phase.py > $O/${T}_dense_kernel_phase_log.txt 2>&1 && python3 tools/phase_report.py $O/ph.bin >> $O/${T}_dense_kernel_phase_log.txt; rm -f $O/ph.bin
[ -x bin/pk_probe ] && bin/pk_probe > $O/${T}_pk_probe.txt 2>&1
[ -x bin/rows_probe ] && timeout -k 5 120 bin/rows_probe combine > $O/${T}_rows_probe_combine.txt 2>&1
SCALE=8 bash tools/soak.sh > /dev/null 2>&1; cp gpurun_out/soak/soak.txt $O/${T}_soak.txt
ls $O
