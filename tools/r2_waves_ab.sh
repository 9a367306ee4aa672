#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/waves; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do for wv in 8 16; do echo "== ICP_NN_WAVES=$wv" >> $O/ab.txt; ICP_NN_WAVES=$wv python3 tools/reg_time.py 6000 >> $O/ab.txt 2>&1; ICP_NN_WAVES=$wv python3 tools/reg_time.py 4000 plane >> $O/ab.txt 2>&1; done; done
ICP_NN_WAVES=16 timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "loop_forms or hall or fuzz or ragged or resident" > $O/pytest_w16.txt 2>&1; tail -3 $O/pytest_w16.txt >> $O/ab.txt
ICP_NN_WAVES=16 ICP_NN_PHASE_PASS=6 ICP_NN_PHASES=$O/ph.bin python3 tools/phase_run.py 9 > /dev/null 2>&1 && python3 tools/phase_report.py $O/ph.bin 16 > $O/phase_w16.txt; python3 tools/phase_sub.py $O/ph.bin 16 >> $O/phase_w16.txt; rm -f $O/ph.bin
cat $O/ab.txt $O/phase_w16.txt
