#!/bin/bash
# like ab_env.sh but tolerant of extra stdout lines (RCCL's banner)
for i in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 100 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('[$e]', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
  done
done
