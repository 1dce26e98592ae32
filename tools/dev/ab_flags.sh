#!/bin/bash
# Interleaved default bench runs under different PGASR_LSTM_FLAGS values on ONE box:  bash tools/dev/ab_flags.sh "0 8192 16384" [rounds]
N=${2:-2}
for i in $(seq 1 $N); do
  for f in $1; do
    PGASR_LSTM_FLAGS=$f timeout -k 10 100 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=$f', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
