#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/phase; mkdir -p $O
for row in 64 128; do
  ICP_NN_ROW=$row ICP_NN_PHASE_PASS=6 ICP_NN_PHASES=$O/ph.bin timeout -k 10 120 python3 tools/phase_run.py 9 > /dev/null 2>&1 && python3 tools/phase_report.py $O/ph.bin > $O/phase_row$row.txt
  [ $row = 64 ] && python3 tools/phase_sub.py $O/ph.bin 8 >> $O/phase_row$row.txt
  rm -f $O/ph.bin
  ICP_NN_ROW=$row timeout -k 10 120 python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1
done
ICP_COMPACT_ROWS=0 timeout -k 10 120 python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1
cat $O/ab.txt $O/phase_row64.txt
