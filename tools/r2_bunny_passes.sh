#!/bin/bash
# Bunny.csv: the time of every pass of a registration inside the kernel (phase log of the last launch of a registration cut after K passes)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/bunny_diag; mkdir -p $O; rm -f $O/passes.txt
NW=${NW:-8}
for K in 1 2 3 4 5 6 8 10 12 14 16 18 20; do
  echo "== pass $K" >> $O/passes.txt
  ICP_NN_PHASES=$O/pp.bin timeout -k 10 120 python3 tools/bunny_phase.py $K > /dev/null 2>&1 && python3 tools/share_report.py $O/pp.bin $NW --brief >> $O/passes.txt 2>&1
done
cat $O/passes.txt
