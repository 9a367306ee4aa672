#!/bin/bash
# round-2 first GPU check: the driver's exact bench command, the default bench, the self-spawn path (two ranks on the one
# GPU), the RCCL leg with one rank
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c1/bench_driver.json 2> gpurun_out/c1/bench_driver.err; echo "rc=$?" >> gpurun_out/c1/bench_driver.err
timeout -k 10 300 python3 bench.py > gpurun_out/c1/bench_default.json 2> gpurun_out/c1/bench_default.err; echo "rc=$?" >> gpurun_out/c1/bench_default.err
ICP_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 400 --warmup 40 > gpurun_out/c1/bench_spawn2.json 2> gpurun_out/c1/bench_spawn2.err; echo "rc=$?" >> gpurun_out/c1/bench_spawn2.err
ICP_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/c1/bench_rccl1.json 2> gpurun_out/c1/bench_rccl1.err; echo "rc=$?" >> gpurun_out/c1/bench_rccl1.err
tail -3 gpurun_out/c1/*.err
