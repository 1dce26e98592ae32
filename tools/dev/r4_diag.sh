#!/bin/bash
# round 4: the six-product kernels with parts of their k-loops switched off (diagnostic build, results invalid)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export PGASR_HIP_LIB=$R/policy_gradient_asr_amd/libpgasr_hip_diag.so
for d in 0 1 4 5 8 13 2 16 18 32 34; do
  echo "== X6_DIAG=$d (1 no W DMA, 2 no MFMA, 4 no A loads, 8 no convert, 16 no frag reads, 32 no barrier)" >> $O/diag.log
  PGASR_X6_DIAG=$d PGASR_TN_DIAG=$(( d==1 ? 1 : (d==2 ? 2 : 0) )) QUICK=1 timeout -k 10 120 python3 tools/dev/tools_gemm6.py 2>&1 | grep "us " >> $O/diag.log || echo failed >> $O/diag.log
done
cat $O/diag.log
