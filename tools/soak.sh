#!/bin/bash
# soak: millions of iterations through the loop forms; any time-out, missing row or differing result ends a run with an error
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; O=gpurun_out/soak; mkdir -p $O; : > $O/soak.txt
run() { echo "== $1" >> $O/soak.txt; shift; env "$@" >> $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt; }
S=${SCALE:-1}
run "hall, resident (default)" timeout -k 10 200 python3 tools/reg_time.py $((4000000 / S))
run "hall, resident, host mailbox" ICP_MAILBOX=host timeout -k 10 200 python3 tools/reg_time.py $((1500000 / S))
run "hall, armed launches" ICP_RESIDENT=0 timeout -k 10 200 python3 tools/reg_time.py $((1000000 / S))
run "hall, point-to-plane" timeout -k 10 200 python3 tools/reg_time.py $((1500000 / S)) plane
run "hall, rows of 128" ICP_NN_ROW=128 timeout -k 10 200 python3 tools/reg_time.py $((1500000 / S))
run "hall, 16 waves per block" ICP_NN_WAVES=16 timeout -k 10 200 python3 tools/reg_time.py $((1500000 / S))
# Bunny.csv (shared rows): the default (armed first registration, resident from the second on), armed throughout, resident from the first pass, the hand-over
for env in "ICP_DEFAULT=1" "ICP_SHARE_AUTO=0" "ICP_RESIDENT=2" "ICP_SHARE_RESIDENT_AFTER=3" "ICP_MAILBOX=host"; do
  run "Bunny.csv, $env" $env timeout -k 10 200 python3 tools/bunny_soak.py $((40 / S))
done
cat $O/soak.txt
