#!/bin/bash
# Bunny.csv, the cold pass (no previous match): length of the find rounds (ICP_NN_PASSES x 8 waves x 64 chunks per round, minima exchanged between rounds)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/bunny_diag; mkdir -p $O; rm -f $O/cold.txt
for P in 8 4 2 1; do
  for K in 1 3; do
  echo "== ICP_NN_PASSES=$P, pass $K" >> $O/cold.txt
  ICP_NN_PASSES=$P ICP_RESIDENT=0 ICP_NN_PHASES=$O/pp.bin timeout -k 10 120 python3 tools/bunny_phase.py $K > /dev/null 2>&1 && python3 tools/share_report.py $O/pp.bin 8 --brief >> $O/cold.txt 2>&1
  done
done
cat $O/cold.txt
