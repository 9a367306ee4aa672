#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/samples; mkdir -p $O; rm -f $O/ab.txt
for g in 256 64 8; do
  echo "== ICP_NN_SAMPLE_GROUPS=$g" >> $O/ab.txt
  ICP_NN_SAMPLE_GROUPS=$g python3 tools/nn_compare.py hall grid128 bunny_res random 2>&1 | grep sparse >> $O/ab.txt
done
cat $O/ab.txt
