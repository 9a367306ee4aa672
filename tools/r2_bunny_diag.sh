#!/bin/bash
# Bunny.csv (35 947^2, rows of 128): where a pass spends its time, pass by pass; env of the caller selects the form
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/bunny_diag; mkdir -p $O; rm -f $O/*.txt
NW=${NW:-8}
for K in ${PASSES:-2 10 20}; do
  echo "== registration cut after $K passes: phase log of the last matching launch" >> $O/phases.txt
  ICP_NN_PHASES=$O/ph$K.bin timeout -k 10 120 python3 tools/bunny_phase.py $K >> $O/phases.txt 2>&1 &&
  python3 tools/phase_report.py $O/ph$K.bin $NW >> $O/phases.txt 2>&1 && python3 tools/share_report.py $O/ph$K.bin $NW >> $O/phases.txt 2>&1
done
cat $O/phases.txt
