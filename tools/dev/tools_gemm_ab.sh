#!/bin/bash
# A/B of the x3w tiles on ONE box: 256 x 256 (default) against the round-1 256 x 128 kernel (PGASR_X3W_TILE=128).
set -e
for rep in 1 2; do
  echo "== 256x256 tile"; python tools/dev/tools_gemm.py 2>&1 | grep -E "x3w"
  echo "== 256x128 tile"; PGASR_X3W_TILE=128 python tools/dev/tools_gemm.py 2>&1 | grep -E "x3w"
done
