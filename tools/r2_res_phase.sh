#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/bunny_diag; mkdir -p $O; rm -f $O/res.txt
for P in ${PASSES:-15}; do
echo "== resident shared, pass $P of the launch" >> $O/res.txt
REPEAT=2 ICP_RESIDENT=2 ICP_NN_PHASE_PASS=$P ICP_NN_PHASES=$O/pr.bin timeout -k 10 120 python3 tools/bunny_phase.py 21 >> $O/res.txt 2>&1 && python3 tools/phase_report.py $O/pr.bin 8 >> $O/res.txt 2>&1 && python3 tools/share_report.py $O/pr.bin 8 >> $O/res.txt 2>&1
done
cat $O/res.txt
