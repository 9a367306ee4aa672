#!/bin/bash
# A/B of the speculative search of resident launches on ONE box: registrations per second, phase log of a steady pass
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/spec; mkdir -p $O
for rep in 1 2; do
  for s in 0 1; do
    echo "== ICP_NN_SPECULATE=$s (run $rep)" >> $O/ab.txt
    ICP_NN_SPECULATE=$s timeout -k 10 120 python3 tools/reg_time.py 4000 >> $O/ab.txt 2>&1
    ICP_NN_SPECULATE=$s timeout -k 10 120 python3 tools/reg_time.py 4000 plane >> $O/ab.txt 2>&1
  done
done
for s in 0 1; do
  ICP_NN_SPECULATE=$s ICP_NN_PHASE_PASS=6 ICP_NN_PHASES=$O/ph$s.bin timeout -k 10 120 python3 tools/phase_run.py 9 > /dev/null 2>&1 && python3 tools/phase_report.py $O/ph$s.bin > $O/phase_spec$s.txt; rm -f $O/ph$s.bin
done
ICP_TRACE=2 timeout -k 10 120 python3 tools/phase_run.py 9 2> $O/trace_spec1.txt > /dev/null
ICP_NN_SPECULATE=0 ICP_TRACE=2 timeout -k 10 120 python3 tools/phase_run.py 9 2> $O/trace_spec0.txt > /dev/null
cat $O/ab.txt
