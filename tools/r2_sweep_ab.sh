#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/sweep; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2 3; do for s in 1 0; do echo "ICP_ROW_SWEEP=$s" >> $O/ab.txt; ICP_ROW_SWEEP=$s python3 tools/reg_time.py 6000 >> $O/ab.txt 2>&1; done; done
cat $O/ab.txt
