#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== default (x3w 128 tile, TN 256 tile)";  CHECK=1 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w 256 tile";                          WHICH=nt,nn PGASR_X3W_TILE=256 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w 256 tile, no DMA";                  WHICH=nt,nn PGASR_X3W_TILE=256 PGASR_X3W_DIAG=1 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w 256 tile, neither";                 WHICH=nt,nn PGASR_X3W_TILE=256 PGASR_X3W_DIAG=3 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== TN 128 tile";                           WHICH=tn PGASR_TN_TILE=128 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
