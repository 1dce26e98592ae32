#!/bin/bash
# round 5: s_setprio on the sweeps' compute waves (they share SIMD 0 / 1 with the loader and the storer wave) by library variant
# (make variant NAME=prio3 DEFS=-DPGASR_SWEEP_PRIO=3), inside the f32 train step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/sweep_prio.log
for lib in libpgasr_hip.so libpgasr_hip_prio3.so libpgasr_hip_prio1.so libpgasr_hip.so libpgasr_hip_prio3.so; do
  [ -f $R/policy_gradient_asr_amd/$lib ] || continue
  echo "== $lib" >> $O/sweep_prio.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/sweep_prio.log
done
cat $O/sweep_prio.log
