#!/bin/bash
# round 5: the time-slab schedule of the streamed weight-gradient products (common.h, pgasr_wslab_next) inside the f32 step.
# Variants built with `make variant` (growth NUM/DEN, cap): s54 = 5/4 cap 176, s54c128, s54c96, s118c64 = 11/8 cap 64; the product is 11/8 cap 168.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/slab_sched.log
for lib in libpgasr_hip.so libpgasr_hip_s54.so libpgasr_hip_s54c128.so libpgasr_hip_s54c96.so libpgasr_hip_s118c64.so libpgasr_hip.so; do
  [ -f $R/policy_gradient_asr_amd/$lib ] || continue
  echo "== $lib" >> $O/slab_sched.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=${PREC:-f32} FLAGS=0 STEPS=40 timeout -k 10 200 python3 tools/dev/r5_instep_diag.py 2>&1 | grep flags >> $O/slab_sched.log
done
cat $O/slab_sched.log
