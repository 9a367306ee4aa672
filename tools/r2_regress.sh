#!/bin/bash
# after a change to nn_match_sparse: the other size classes it serves (hall rows of 128, a 65 k / 131 k cloud, the 10 M-point share)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/regress; mkdir -p $O; rm -f $O/r.txt
# (ICP_LIB_PATH=ab/<older build>.so bash tools/r2_regress.sh: the same measurements with another build)
echo "library: ${ICP_LIB_PATH:-this build}" >> $O/r.txt
for row in 64 128; do echo "== hall ICP_NN_ROW=$row" >> $O/r.txt; ICP_NN_ROW=$row timeout -k 10 200 python3 tools/reg_time.py 4000 >> $O/r.txt 2>&1; done
echo "== Bunny.csv" >> $O/r.txt; timeout -k 10 200 python3 tools/bunny_time.py >> $O/r.txt 2>&1
echo "== configs[4] share of one rank of 8 (bench --config s5)" >> $O/r.txt; timeout -k 10 400 python3 bench.py --config s5 --no-cpu-baseline --steps 10 --warmup 2 >> $O/r.txt 2>> $O/s5.err
echo "== grids" >> $O/r.txt; timeout -k 10 300 python3 tools/grid_time.py >> $O/r.txt 2>&1
cat $O/r.txt
