#!/bin/bash
# round 4: K = 512 feeds in quarters (first tiles) against whole tiles: f32 step phases; then the bit-identity tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/quarters.log
for rep in 1 2; do for q in 512 1024; do
  echo "== PGASR_X6_QUARTER_K=$q" >> $O/quarters.log
  PGASR_X6_QUARTER_K=$q PREC=f32 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/quarters.log
done; done
python3 - <<PY
import json
for l in open("$O/quarters.log"):
    if l.startswith("=="): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    p=d["phases"]; print(f"   {d['ms_per_step']:.2f} ms  front {p['front_end']:.2f} fwd {p['forward_sweeps']:.2f} loss {p['loss_section']:.2f} bwd {p['backward_sweeps']:.2f} tail {p['tail']:.2f}  sweeps " + " ".join(f"{x:.2f}" for x in p["sweeps_in_launch_order"]))
PY
timeout -k 10 500 python3 -m pytest tests/test_dense_lstm_gpu.py tests/test_train_step_gpu.py -x -q -k "fed_by or f32_mode or feed_ahead or streamed_weight or x6w" > $O/t3.log 2>&1; echo "rc=$?" >> $O/t3.log; tail -n 5 $O/t3.log
