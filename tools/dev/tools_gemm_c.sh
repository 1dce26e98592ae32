#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== x3w c256 (8 waves, cooperative split)";  WHICH=nt,nn PGASR_X3W_TILE=c python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w 256 tile (4 waves)";                  WHICH=nt,nn PGASR_X3W_TILE=256 python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w 128 tile";                            WHICH=nt,nn python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
echo "== x3w c256 again";                          WHICH=nt,nn PGASR_X3W_TILE=c python tools/dev/tools_gemm3.py 2>&1 | grep -v amdgpu.ids
