#!/bin/bash
# Packed (pre-tiled) three-plane W operand of the six-product kernel against row-major planes: tests, stand-alone GEMM, f32 step A/B/A/B.
set -e
mkdir -p gpurun_out/r4
python -m pytest tests/test_dense_lstm_gpu.py -x -q -k "x6w or fed_by or feed" > gpurun_out/r4/packed_tests.txt 2>&1 || { tail -n 30 gpurun_out/r4/packed_tests.txt; exit 1; }
tail -n 2 gpurun_out/r4/packed_tests.txt
QUICK=1 python tools/dev/tools_gemm6.py 2>&1 | tee gpurun_out/r4/packed_gemm6.txt
for rep in 1 2; do
  for p in 0 1; do
    echo "== PGASR_X6_PACKED=$p (rep $rep)"
    PGASR_X6_PACKED=$p PREC=f32 STEPS=60 python tools/dev/tools_precision_phases.py 2>&1 | grep -E "ms per step|ms/step|gemm_feed_x6c|gemm_x6c|fwd|bwd" | head -n 14
  done
done 2>&1 | tee gpurun_out/r4/packed_step.txt
