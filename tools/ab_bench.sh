#!/bin/bash
# same-box A/B of bench.py configs between an older build and the tree's:  OLD=ab/libicp_xxx.so CFGS="s5 bunny" bash tools/ab_bench.sh
# (the boxes of the pool differ by up to 25 %: only runs of ONE gpurun call compare)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
OLD=${OLD:-ab/libicp_r2_final.so}
for c in ${CFGS:-s5}; do
  for rep in 1 2; do
    for lib in old new; do
      if [ $lib = old ]; then export ICP_LIB_PATH=$R/$OLD; else unset ICP_LIB_PATH; fi
      python3 bench.py --leg main --config $c --no-cpu-baseline ${ARGS} 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$c $lib rep $rep: value %.5g it/s  ms/step %.5g  kernel avg %.2f us  frac %.4f  hits_box %d hits_full %d' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r['executed']['work_counters_one_registration']['hits_box'], r['executed']['work_counters_one_registration']['hits_full']))"
    done
  done
done
