#!/bin/bash
# round 4 close-out on one box: full GPU suite, smoke, the default bench line (f32) and the bf16x3 one, kernel stats of both
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/full2.log 2>&1; echo "pytest rc=$?" >> $O/full2.log; tail -n 4 $O/full2.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
timeout -k 10 400 python3 bench.py > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?"
timeout -k 10 400 python3 bench.py --precision bf16x3 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; echo "bench bf16x3 rc=$?"
python3 -c "
import json
for f in ('bench_f32','bench_bf16x3'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), round(d['ms_per_step'],3), d['long_run']['ms_per_step'], d['roofline']['floor_frac'], {k: (round(v['avg_launch_us'],1), round(v['frac_of_2500'],3)) for k,v in d['roofline_gemm']['kernels'].items()})
"
TAG=r04; RUN=$(date +%H%M%S); P=$R/gpurun_out/${TAG}_prof_$RUN; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for prec in f32 bf16x3; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/${TAG}_stats_$prec" -- python3 "$R/bench.py" --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > "$P/stats_$prec.log" 2>&1 || echo "stats pass $prec failed"
  PREC=$prec timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/${TAG}_fedfetch_$prec" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$P/fedfetch_$prec.log" 2>&1 || echo "fed fetch pass $prec failed"
  PREC=$prec timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/${TAG}_fedwrite_$prec" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$P/fedwrite_$prec.log" 2>&1 || echo "fed write pass $prec failed"
done
echo final done
