#!/bin/bash
# round 4: ping-pong variants of the six-product kernels (PGASR_X6_VAR 3/4, PGASR_T6_VAR 1/2) against the staggered ones
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for rep in 1 2; do
for v in "0 0" "3 1" "4 2" "1 0"; do
  set -- $v
  echo "== X6_VAR=$1 T6_VAR=$2" >> $O/pp.log
  PGASR_X6_VAR=$1 PGASR_T6_VAR=$2 QUICK=1 timeout -k 10 120 python3 tools/dev/tools_gemm6.py 2>&1 | grep -v "amdgpu.ids\|issued bf16" >> $O/pp.log || echo failed >> $O/pp.log
done
done
cat $O/pp.log
