#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by several per cent): interleaved default bench runs.
#   bash tools/dev/ab_bench.sh policy_gradient_asr_amd/libpgasr_hip_old.so [rounds]
OLD=$1; N=${2:-3}
for i in $(seq 1 $N); do
  for lib in "$OLD" ""; do
    if [ -n "$lib" ]; then export PGASR_HIP_LIB=$PWD/$lib; tag=old; else unset PGASR_HIP_LIB; tag=new; fi
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
