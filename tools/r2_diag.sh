#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/diag; mkdir -p $O
for row in 64 128; do
  ICP_NN_ROW=$row timeout -k 10 120 python3 tools/r2_diag.py > $O/work_row$row.json 2> $O/work_row$row.err
  ICP_NN_ROW=$row ICP_TRACE=2 timeout -k 10 120 python3 tools/reg_time.py 26 2> $O/trace_row$row.txt > /dev/null
  ICP_NN_ROW=$row timeout -k 10 120 python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1
done
cat $O/ab.txt; cat $O/work_row64.json; cat $O/work_row128.json
