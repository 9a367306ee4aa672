#!/bin/bash
# SQ counters of the hierarchical kernel on the 10 M-point share (early passes: is the VALU the limit?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq_s5; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/s5_time.py 0 8 4 > $O/$tag.log 2>&1 || echo "pmc $set exit $?"
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_sq_s5"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "nn_match_sparse<1" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,v in acc.items():
        print(k, file=out)
        for c,vals in sorted(v.items()):
            print(f"  {c:28s} n={len(vals):4d} mean {sum(vals)/len(vals):16.1f}  last {vals[-1]:16.1f}", file=out)
print(open(O+"/summary.txt").read())
PY
rm -rf $O/SQ_* $O/GRBM_*
