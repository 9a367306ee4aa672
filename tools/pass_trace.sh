#!/bin/bash
# duration of every matching launch of one bench leg, in launch order (rocprofv3 kernel trace):  CFG=s5 ARGS="--steps 30 --warmup 0" bash tools/pass_trace.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pass_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --leg main --config ${CFG:-s5} --no-cpu-baseline ${ARGS:---steps 30 --warmup 0} > $O/line.json 2> $O/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void icp::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "nn_match_" in r["Kernel_Name"]]
cur, out = None, []
for n, us in seq:
    if n != cur:
        if out: print(cur, "x", len(out), ":", " ".join(f"{u:.0f}" for u in out))
        cur, out = n, []
    out.append(us)
if out: print(cur, "x", len(out), ":", " ".join(f"{u:.0f}" for u in out))
PY
rm -rf $O/kt
