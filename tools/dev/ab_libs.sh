#!/bin/bash
# Interleaved stand-alone sweep timings + default bench for several builds of the library on ONE box:
#   bash tools/dev/ab_libs.sh "libpgasr_hip.so libpgasr_hip_v_X.so ..."
for i in 1 2; do
  for lib in $1; do
    export PGASR_HIP_LIB=$PWD/policy_gradient_asr_amd/$lib
    a=$(python tools/dev/tools_sweep_once.py 2>&1 | tail -1)
    b=$(timeout -k 10 100 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items() if 'lstm' in k})")
    echo "$lib | $a | $b"
  done
done
