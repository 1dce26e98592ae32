#!/bin/bash
# round 5: how many time-ordered tile groups at the head of an input-gradient FEED (K = 2048, six-product kernel) are split into K-quarters.
# A whole 256 x 256 x 2048 tile takes a CU ~220 us; with 16 groups (32 tiles, round 4) the fed backward sweep runs out of rows ~150 us after
# its start and waits for the second wave of whole tiles (tools/dev/r5_ring_wait.py: ~0.25 ms of waiting per fed backward sweep).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/split_groups.log
for gsz in 16 32 48 64 16 64; do
  echo "== PGASR_X6_SPLIT_GROUPS=$gsz" >> $O/split_groups.log
  PGASR_X6_SPLIT_GROUPS=$gsz FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/split_groups.log
done
cat $O/split_groups.log
