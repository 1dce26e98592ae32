#!/bin/bash
# K = 512 projections in quarters again (packed W planes now), and fewer split tiles; f32 step, same box
for cfg in "1024 32" "512 32" "512 16" "512 8" "1024 16" "1024 32"; do
  set -- $cfg
  echo "== PGASR_X6_QUARTER_K=$1 PGASR_X6_SPLIT_MAX=$2"
  PGASR_X6_QUARTER_K=$1 PGASR_X6_SPLIT_MAX=$2 PREC=f32 STEPS=60 python tools/dev/tools_precision_phases.py 2>&1 | grep ms_per_step | sed 's/"instrumented.*//' | cut -c1-420
done
