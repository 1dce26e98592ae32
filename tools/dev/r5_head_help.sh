#!/bin/bash
# round 5: the K-split head items of an input-gradient feed also on the idle CUs of the sweep's own XCDs (PGASR_X6_HEAD_HELP=1) or not (0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/head_help.log
for cfg in "0 16" "1 16" "1 32" "1 64" "0 16" "1 16"; do
  set -- $cfg
  echo "== PGASR_X6_HEAD_HELP=$1 PGASR_X6_SPLIT_GROUPS=$2" >> $O/head_help.log
  PGASR_X6_HEAD_HELP=$1 PGASR_X6_SPLIT_GROUPS=$2 FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/head_help.log
done
cat $O/head_help.log
