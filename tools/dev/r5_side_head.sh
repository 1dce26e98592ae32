#!/bin/bash
# round 5 (the wiring this script switched was removed again after the measurement; kept for the record): the head on a stream of its own (PGASR_X6_SIDE_HEAD=1;
# queue words zeroed with the step's counters) against the single launch behind the consuming sweep's registration (0), inside the f32 train step.
# "h f" = PGASR_X6_SIDE_HEAD PGASR_X6_FWD_SPLIT_GROUPS
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/side_head.log
for c in "0 2" "1 2" "1 6" "0 2" "1 2" "1 4"; do
  set -- $c
  echo "== PGASR_X6_SIDE_HEAD=$1 PGASR_X6_FWD_SPLIT_GROUPS=$2" >> $O/side_head.log
  PGASR_X6_SIDE_HEAD=$1 PGASR_X6_FWD_SPLIT_GROUPS=$2 FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep "flags\|Error\|error" >> $O/side_head.log
done
cat $O/side_head.log
