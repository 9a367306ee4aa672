#!/bin/bash
# XCD-aware rows in the 64-point-row kernel: HBM traffic of a stand-alone pass (PMC) and the hall registrations, this build against ab/libicp_head.so
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/xcd; mkdir -p $O; rm -rf $O/*
for lib in new head; do
  if [ $lib = head ]; then export ICP_LIB_PATH=$GRAFT_REPO_ROOT/ab/libicp_head.so; else unset ICP_LIB_PATH; fi
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf_$lib -- python3 $GRAFT_REPO_ROOT/tools/nn_only.py 30 > $O/pf_$lib.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw_$lib -- python3 $GRAFT_REPO_ROOT/tools/nn_only.py 30 > $O/pw_$lib.log 2>&1 )
  echo "== $lib" >> $O/ab.txt
  python3 tools/pmc_summary.py $O/pf_$lib $O/pw_$lib $O/traffic_$lib.json 2>&1 | grep "nn_match_row64" >> $O/ab.txt
  for i in 1 2; do python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1; done
  python3 tools/reg_time.py 4000 plane >> $O/ab.txt 2>&1
  python3 tools/nn_compare.py hall grid128 2>&1 | grep sparse >> $O/ab.txt
done
cat $O/ab.txt
