#!/bin/bash
# round 5: do the gated weight-gradient workgroups (persistent, 108 KB of LDS each, waiting for slabs) keep the CUs from the GEMM that FEEDS
# the sweep?  PGASR_T6_GRID = persistent workgroups of their masked pass (256 = one per CU; half leave at once on the sweep's XCDs).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/t6_grid.log
for gsz in 256 192 128 96 64 256; do
  echo "== PGASR_T6_GRID=$gsz" >> $O/t6_grid.log
  PGASR_T6_GRID=$gsz FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/t6_grid.log
done
cat $O/t6_grid.log
