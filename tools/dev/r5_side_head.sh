#!/bin/bash
# round 5: the six-product feeds' K-split head as a launch of its own on the feeding stream, right behind the previous sweep (PGASR_X6_SIDE_HEAD=1)
# against the single launch behind the consuming sweep's registration (0), inside the f32 train step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/side_head.log
for v in 0 1 0 1; do
  echo "== PGASR_X6_SIDE_HEAD=$v" >> $O/side_head.log
  PGASR_X6_SIDE_HEAD=$v FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/side_head.log
done
cat $O/side_head.log
