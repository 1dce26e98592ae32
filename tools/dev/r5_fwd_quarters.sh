#!/bin/bash
# round 5: K = 512 projections (forward feeds) with only the FIRST few tile groups split into K-quarters (round 4 measured 16 groups: a loss).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/fwd_quarters.log
for cfg in "1024 16" "512 1" "512 2" "512 4" "512 8" "1024 16"; do
  set -- $cfg
  echo "== PGASR_X6_QUARTER_K=$1 PGASR_X6_SPLIT_GROUPS=$2" >> $O/fwd_quarters.log
  PGASR_X6_QUARTER_K=$1 PGASR_X6_SPLIT_GROUPS=$2 FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/fwd_quarters.log
done
cat $O/fwd_quarters.log
