#!/bin/bash
# round 5: a thinner forward feed (fewer persistent workgroups of its masked pass, PGASR_X6_FWD_FEED_GRID) -- does the sweep beside it keep more clock?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/fwd_feed_grid.log
for gq in 256 192 128 96 64 256; do
  echo "== PGASR_X6_FWD_FEED_GRID=$gq" >> $O/fwd_feed_grid.log
  PGASR_X6_FWD_FEED_GRID=$gq FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/fwd_feed_grid.log
done
cat $O/fwd_feed_grid.log
