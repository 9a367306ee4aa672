#!/bin/bash
# round 2 extras for profiles/: two ranks on the one GPU through the self-spawn path, the RCCL leg with one rank, SQ counters
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/extra; mkdir -p $O
ICP_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 > $O/bench_2ranks_on_1gpu_rehearsal.json 2> $O/spawn2.err; echo "rc=$?" >> $O/spawn2.err
ICP_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_rccl_leg_1rank.json 2> $O/rccl1.err; echo "rc=$?" >> $O/rccl1.err
bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq/summary.txt $O/pmc_sq_counters_matching_kernel.txt
tail -2 $O/*.err; cat $O/pmc_sq_counters_matching_kernel.txt
