#!/bin/bash
# soak: millions of iterations through the loop forms; any time-out or missing row ends a run with an error
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/soak; mkdir -p $O; rm -f $O/soak.txt
echo "== resident (default)" >> $O/soak.txt; timeout -k 10 200 python3 tools/reg_time.py 4000000 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
echo "== resident, host mailbox" >> $O/soak.txt; ICP_MAILBOX=host timeout -k 10 200 python3 tools/reg_time.py 1500000 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
echo "== armed launches" >> $O/soak.txt; ICP_RESIDENT=0 timeout -k 10 200 python3 tools/reg_time.py 1000000 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
echo "== point-to-plane" >> $O/soak.txt; timeout -k 10 200 python3 tools/reg_time.py 1500000 plane >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
echo "== rows of 128" >> $O/soak.txt; ICP_NN_ROW=128 timeout -k 10 200 python3 tools/reg_time.py 1500000 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
echo "== 16 waves per block" >> $O/soak.txt; ICP_NN_WAVES=16 timeout -k 10 200 python3 tools/reg_time.py 1500000 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
cat $O/soak.txt
# Bunny.csv (shared rows): armed launches, a resident kernel sharing its rows, and the hand-over between the two
for env in "ICP_DEFAULT=1" "ICP_RESIDENT=2" "ICP_SHARE_RESIDENT_AFTER=3" "ICP_MAILBOX=host"; do
  echo "== Bunny.csv, $env" >> $O/soak.txt; env $env timeout -k 10 200 python3 tools/bunny_soak.py 40 >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt
done
tail -12 $O/soak.txt
