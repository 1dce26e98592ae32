#!/bin/bash
# round 5, after the 16-byte slab accesses: the K = 512 forward projections' first tile groups in K-quarters (PGASR_X6_FWD_SPLIT_GROUPS; 0 = whole
# tiles only), inside the f32 train step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/fwd_quarters2.log
for gq in 0 2 4 8 16 0 4; do
  echo "== PGASR_X6_FWD_SPLIT_GROUPS=$gq" >> $O/fwd_quarters2.log
  PGASR_X6_FWD_SPLIT_GROUPS=$gq FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/fwd_quarters2.log
done
cat $O/fwd_quarters2.log
