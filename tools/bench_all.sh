#!/bin/bash
# every BASELINE config through bench.py on the GPU box:  OUT=gpurun_out/<dir> bash tools/bench_all.sh [configs...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${OUT:-gpurun_out/bench_all}
mkdir -p $O
cd $R
CFGS=${@:-hall hall_plane bunny s5 cpu_f64}
for c in $CFGS; do
  echo "== $c"
  timeout -k 10 900 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || { echo "bench $c exit $?"; tail -5 $O/bench_$c.err; }
  python3 - $O/bench_$c.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, c = d.get("roofline", {}), d.get("cpu_baseline", {})
    print(f"  value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.5g}  frac {r.get('frac')}  achieved {r.get('achieved')}  avg_launch_us {r.get('avg_launch_us')} ppl {r.get('passes_per_launch')}  cpu {c.get('value')} ({c.get('cores')} core)")
except Exception as e:
    print("  no line:", e)
PY
done
