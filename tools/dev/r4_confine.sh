#!/bin/bash
# round 4: the f32 step with the GEMMs beside the sweeps confined to the free XCDs or not (they are GEMM-bound in this mode)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/confine.log
for c in "1 1" "0 1" "1 0" "0 0"; do set -- $c
  for v in ${VARS:-0}; do
    echo "== CONFINE=$1 CONFINE_FEED=$2 X6_VAR=$v PREC=${PREC:-f32}" >> $O/confine.log
    PGASR_CONFINE=$1 PGASR_CONFINE_FEED=$2 PGASR_X6_VAR=$v PREC=${PREC:-f32} STEPS=30 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/confine.log || echo failed >> $O/confine.log
  done
done
python3 - <<PY
import json
for l in open("$O/confine.log"):
    if l.startswith("=="): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    p=d["phases"]; print(f"   {d['ms_per_step']:.2f} ms  front {p['front_end']:.2f} fwd {p['forward_sweeps']:.2f} loss {p['loss_section']:.2f} bwd {p['backward_sweeps']:.2f} tail {p['tail']:.2f}  sweeps " + " ".join(f"{x:.2f}" for x in p["sweeps_in_launch_order"]))
PY
