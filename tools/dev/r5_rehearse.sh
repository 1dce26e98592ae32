#!/bin/bash
# round 5 (re-run of the round-4 recipe): the N > 1 bench path on ONE GPU (two ranks on cuda:0, gloo collectives staged through the host): plumbing only -- NOT a scaling number
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
PGASR_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 10 --warmup 3 --long-steps 20 --no-cpu-baseline > $O/rehearse.json 2> $O/rehearse.err
echo "rehearse rc=$?"
tail -n 5 $O/rehearse.err
python3 -c "
import json; d=json.load(open('$O/rehearse.json')); print({k: d.get(k) for k in ('n_gpus','rccl_world_size','collective_backend','ms_per_step','ms_per_step_per_rank')}); print(d.get('long_run')); print(d.get('inputs_resident')); print(d.get('bucketed'))"
