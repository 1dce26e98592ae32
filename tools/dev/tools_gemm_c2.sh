#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== c256";                      WHICH=nt,nn PGASR_X3W_TILE=c python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== c256, no memory waits";     WHICH=nt,nn PGASR_X3W_TILE=c PGASR_X3W_DIAG=1 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== c256, no MFMA";             WHICH=nt,nn PGASR_X3W_TILE=c PGASR_X3W_DIAG=2 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== c256, neither";             WHICH=nt,nn PGASR_X3W_TILE=c PGASR_X3W_DIAG=3 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
