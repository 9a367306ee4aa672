#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/specfloor; mkdir -p $O; rm -f $O/ab.txt
for cfg in "2.0 1e-3" "2.0 1e-4" "2.0 1e-5" "1.5 1e-4" "1.0 1e-4" "3.0 1e-4"; do
  set -- $cfg
  echo "== gain $1 floor $2" >> $O/ab.txt
  ICP_SPEC_GAIN=$1 ICP_SPEC_FLOOR=$2 python3 tools/r2_diag.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['whole']; print('lists',d['spec_lists'],'covered',d['spec_covered'],'hits per used list %.1f' % (d['spec_hits']/max(1,d['spec_covered'])),'ordinary hits',d['list_hits'])" >> $O/ab.txt
  ICP_SPEC_GAIN=$1 ICP_SPEC_FLOOR=$2 python3 tools/reg_time.py 6000 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
