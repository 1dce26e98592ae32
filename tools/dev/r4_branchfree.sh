#!/bin/bash
# Publish stores of the sweeps branch-free (two descriptors, one with an empty range) against if / else around every store: library A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_dense_lstm_gpu.py -x -q -k "blstm or streamed or sweep_error or write_through" > $O/bf_tests.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/bf_tests.log
for rep in 1 2; do for lib in libpgasr_hip_branchy.so libpgasr_hip.so; do
  echo "== $lib (rep $rep)"
  for p in bf16x3 f32; do PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=$p timeout -k 10 120 python3 tools/dev/tools_sweep_once.py 2>&1 | grep flags; done
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 STEPS=60 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep ms_per_step | sed 's/"instrumented.*//' | cut -c1-420
done; done 2>&1 | tee $O/branchfree.txt
