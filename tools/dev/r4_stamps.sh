#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/x6stamps.log
for v in ${VARS:-0 4 5 6}; do for d in ${DIAGS:-0}; do
  PGASR_X6_VAR=$v PGASR_X6_DIAG=$d timeout -k 10 60 python3 tools/dev/tools_x6_stamps.py 2>&1 | grep -v "amdgpu.ids\|segments:" >> $O/x6stamps.log
done; done
cat $O/x6stamps.log
for v in ${VARS:-0 4 5 6}; do echo "== X6_VAR=$v"; PGASR_X6_VAR=$v QUICK=1 timeout -k 10 120 python3 tools/dev/tools_gemm6.py 2>&1 | grep "us "; done
