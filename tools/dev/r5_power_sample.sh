#!/bin/bash
# round 5: what rocm-smi says about power and clocks while the f32 train step runs (200 + 2000 steps in the background, a sample every 0.5 s),
# then while the bare-MFMA kernel runs, then idle.  Supports NOTES 0.46: the sweeps' in-step pace is the clock the GEMMs leave them.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
L=$O/power_sample.log; rm -f $L
sample() { for i in $(seq 1 $1); do echo "-- $2 sample $i" >> $L; timeout 10 rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" >> $L; sleep 0.5; done; }
sample 2 idle
python3 bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > $O/power_bench.json 2> $O/power_bench.err &
BP=$!
sleep 6
sample 6 "f32 train step"
wait $BP
if [ -x tools/mfma_peak.bin ]; then (for i in 1 2 3; do tools/mfma_peak.bin > /dev/null 2>&1; done) & MP=$!; sleep 0.3; sample 3 "bare MFMA"; wait $MP; fi
sample 1 idle
cat $L | head -120
