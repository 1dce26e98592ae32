#!/bin/bash
# round 5: the beam kernel's pop rounds as a rolled loop (product) against sixteen rounds written out (libpgasr_hip_unroll.so = -DPGASR_BEAM_UNROLL=1), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
rm -f $O/beam_unroll.log
for lib in libpgasr_hip.so libpgasr_hip_unroll.so libpgasr_hip.so libpgasr_hip_unroll.so; do
  echo "== $lib" >> $O/beam_unroll.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib timeout -k 10 100 python3 tools/dev/r5_beam_var.py 2>&1 | grep "il,collapse" >> $O/beam_unroll.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib timeout -k 10 100 python3 tools/dev/tools_beam_time.py 2>&1 | grep small >> $O/beam_unroll.log
done
cat $O/beam_unroll.log
