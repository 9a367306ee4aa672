#!/bin/bash
# the 10 M-point share (rank 0 of 8): duration of every matching launch of a 30-iteration registration (kernel trace)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s5trace; mkdir -p $O; rm -rf $O/kt
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $GRAFT_REPO_ROOT/tools/s5_time.py 0 8 30 > $O/run.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/s5trace/kt/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "nn_match_sparse" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "launches of", rows[-1]["Kernel_Name"][:60])
print("durations (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-31:]])
PY
