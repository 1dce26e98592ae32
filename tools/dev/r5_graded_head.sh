#!/bin/bash
# round 5: the graded head of the six-product feed GEMMs (gemm_x6.hip): tile groups in K-eighths / quarters / halves in front of the whole tiles,
# inside the f32 train step.  "e q h" = PGASR_X6_SPLIT8_GROUPS PGASR_X6_SPLIT_GROUPS PGASR_X6_SPLIT2_GROUPS.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/graded_head.log
COMBOS=${COMBOS:-"0,16,0 0,16,16 0,16,32 2,14,16 2,14,32 4,12,32 0,16,0"}
for c in $COMBOS; do
  IFS=, read e q h <<< "$c"
  echo "== eighths $e quarters $q halves $h" >> $O/graded_head.log
  PGASR_X6_SPLIT8_GROUPS=$e PGASR_X6_SPLIT_GROUPS=$q PGASR_X6_SPLIT2_GROUPS=$h FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/graded_head.log
done
cat $O/graded_head.log
