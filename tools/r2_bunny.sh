#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/bunny; mkdir -p $O; rm -f $O/ab.txt
for row in 128 64; do
  echo "== ICP_NN_ROW=$row" >> $O/ab.txt
  ICP_NN_ROW=$row timeout -k 10 200 python3 tools/bunny_time.py >> $O/ab.txt 2>&1
  ICP_NN_ROW=$row timeout -k 10 200 python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
