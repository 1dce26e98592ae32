#!/bin/bash
# The fused head kernel in the train step, A/B/A/B on one box (PGASR_FUSED_HEAD=0: general GEMM + log-softmax pass).
set -e
mkdir -p gpurun_out/r4
for rep in 1 2; do for h in 0 1; do for p in f32 bf16x3; do
  echo "== fused head $h, $p (rep $rep)"; PGASR_FUSED_HEAD=$h PREC=$p STEPS=60 python tools/dev/tools_precision_phases.py 2>&1 | grep ms_per_step | cut -c1-420
done; done; done | tee gpurun_out/r4/head_step.txt
