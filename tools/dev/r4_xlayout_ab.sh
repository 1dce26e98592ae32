#!/bin/bash
# round 4: forward exchange layout [kc][plane][n] (default) against the old [kc][n][plane] build (libpgasr_hip_oldx.so): tests, stand-alone sweeps, stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests/test_dense_lstm_gpu.py -x -q -k "blstm or encoder or seq2seq" > $O/xl_tests.log 2>&1; echo "pytest rc=$?" >> $O/xl_tests.log
tail -n 4 $O/xl_tests.log
for rep in 1 2; do
for lib in libpgasr_hip.so libpgasr_hip_oldx.so; do
  for p in bf16x3 f32; do
    echo -n "$lib " >> $O/xl.log
    PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=$p timeout -k 10 120 python3 tools/dev/tools_sweep_once.py 2>&1 | grep flags >> $O/xl.log
  done
done
done
for p in bf16x3 f32; do echo "== stamps $p (new layout)" >> $O/xl.log; PREC=$p timeout -k 10 120 python3 tools/dev/tools_stamps.py 2>&1 | grep -v "^\[\[\|se[0-9]\|distinct\|XCC id\|CU id\|amdgpu.ids" >> $O/xl.log; done
cat $O/xl.log
