#!/bin/bash
# One parametrised driver for same-box comparisons on the GPU box (the boxes of the pool differ by up to 25 %: only the runs of
# ONE gpurun call compare).  Every command of CMDS runs once under every environment of VARIANTS; everything goes to $OUT.
#   VARIANTS="ICP_NN_ROW=64;ICP_NN_ROW=128"  CMDS="python3 tools/reg_time.py 4000;python3 tools/bunny_time.py"  bash tools/ab.sh
#   VARIANTS=";ICP_LIB_PATH=$PWD/ab/libicp_r2_final.so" CMDS="python3 tools/reg_time.py 4000" bash tools/ab.sh     (a build against another)
#   PHASES=1: every command runs with ICP_NN_PHASES set and tools/phase_report.py (+ share_report.py, NW waves per block) follows it
# (replaces round 2's r2_*.sh one-offs: r2_bunny, r2_diag, r2_phase, r2_regress, r2_share_ab, r2_spec_ab, r2_waves_ab, r2_xcd_ab, ...)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=${OUT:-gpurun_out/ab/ab.txt}; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
IFS=';' read -ra VARR <<< "${VARIANTS:-;}"
IFS=';' read -ra CARR <<< "${CMDS:-python3 tools/reg_time.py 4000}"
[ ${#VARR[@]} -eq 0 ] && VARR=("")
for v in "${VARR[@]}"; do
  for cmd in "${CARR[@]}"; do
    echo "== [${v:-default}] $cmd" >> "$OUT"
    if [ -n "$PHASES" ]; then
      ph=$(dirname "$OUT")/ph.bin
      env $v ICP_NN_PHASES=$ph timeout -k 10 ${TIMEOUT:-300} $cmd >> "$OUT" 2>&1 && python3 tools/phase_report.py $ph ${NW:-} >> "$OUT" 2>&1
      [ -n "$SHARE_REPORT" ] && python3 tools/share_report.py $ph ${NW:-8} >> "$OUT" 2>&1
      rm -f $ph
    else
      env $v timeout -k 10 ${TIMEOUT:-300} $cmd >> "$OUT" 2>&1 || echo "FAILED rc=$?" >> "$OUT"
    fi
  done
done
cat "$OUT"
