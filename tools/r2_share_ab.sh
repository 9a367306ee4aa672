#!/bin/bash
# Bunny.csv: rows of 128 as 16-wave blocks (round 1/2 so far) against 8-wave blocks, resident / armed, rows shared or not
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/share; mkdir -p $O; rm -f $O/ab.txt
run() { echo "== $*" >> $O/ab.txt; env "$@" timeout -k 10 200 python3 tools/bunny_time.py >> $O/ab.txt 2>&1 || echo "FAILED rc=$?" >> $O/ab.txt; }
for v in ${VARIANTS:-a f c g h}; do
case $v in
a) run ICP_NN_WAVES128=16 ;;
b) run ICP_RESIDENT=2 ICP_NN_SHARE_RESIDENT=0 ;;
c) run ICP_SHARE_RESIDENT_AFTER=-1 ;;
d) run ICP_RESIDENT=0 ICP_NN_SHARE=0 ;;
e) run ICP_RESIDENT=0 ICP_ARMED=0 ;;
f) run ICP_DEFAULT=1 ;;
g) run ICP_SHARE_RESIDENT_AFTER=3 ;;
h) run ICP_SHARE_RESIDENT_AFTER=10 ;;
i) run ICP_RESIDENT=2 ;;
esac
done
cat $O/ab.txt
