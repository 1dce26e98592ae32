#!/bin/bash
# Interleaved default bench runs under different environment settings on ONE box:  bash tools/dev/ab_env.sh "A=1" "B=0 C=1" ...
for i in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 100 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
