#!/bin/bash
# SQ / cache counters of the matching kernels of one command, one rocprofv3 --pmc pass per counter set (kernel trace only):
#   SETS="A B C;D E" CMD="python3 tools/s5_time.py 0 8 4" FILTER="nn_match_sparse<1" OUT=gpurun_out/pmc bash tools/pmc_sets.sh
R=$GRAFT_REPO_ROOT; O=$R/${OUT:-gpurun_out/pmc_sets}; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SETARR <<< "$SETS"
for set in "${SETARR[@]}"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- ${CMD/tools/$R\/tools} > $O/$tag.log 2>&1 || echo "pmc $set exit $?"
done
FILTER="$FILTER" python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/"+os.environ.get("OUT","gpurun_out/pmc_sets")
flt=os.environ.get("FILTER","nn_match")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if flt in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,v in acc.items():
        print(k, file=out)
        for c,vals in sorted(v.items()):
            print(f"  {c:30s} n={len(vals):4d} mean {sum(vals)/len(vals):18.1f}  last {vals[-1]:18.1f}", file=out)
print(open(O+"/summary.txt").read())
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
