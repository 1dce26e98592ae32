#!/bin/bash
# round 5: what does the loader wave's path cost the sweeps INSIDE the step?  (a) diagnostic bits on the product library: loader waits for
# the staging ring but issues no LDS-DMA (0x2000), LDS-DMA without waiting (0x4000), neither (0x200); (b) deeper loader leads by library:
# _l1 = FWD_LEAD 6 / BWD_LEAD 7, _l2 = 8 / 9 with staging rings of 20 / 14 steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/loader_ab.log
echo "== libpgasr_hip.so diag" >> $O/loader_ab.log
FLAGS=0,0x2000,0x4000,0x200,0 timeout -k 10 300 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/loader_ab.log
for lib in libpgasr_hip_l1.so libpgasr_hip_l2.so libpgasr_hip.so; do
  echo "== $lib" >> $O/loader_ab.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib FLAGS=0,0 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/loader_ab.log
done
cat $O/loader_ab.log
