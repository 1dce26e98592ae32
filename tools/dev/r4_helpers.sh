#!/bin/bash
# round 4: 4 / 6 / 8 helper workgroups per cluster UNDER LOAD (the f32 step; the backward sweep beside a stream copy)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/helpers.log
for lib in libpgasr_hip.so libpgasr_hip_h6.so libpgasr_hip_h8.so; do
  echo "== $lib" >> $O/helpers.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 WGS="0 64" timeout -k 10 120 python3 tools/dev/tools_bwd_beside_stream.py 2>&1 | grep -v amdgpu.ids >> $O/helpers.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/helpers.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=bf16x3 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/helpers.log
done
python3 - <<PY
import json
for l in open("$O/helpers.log"):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    p=d["phases"]; print(f"   {d['precision']} {d['ms_per_step']:.2f} ms  front {p['front_end']:.2f} fwd {p['forward_sweeps']:.2f} loss {p['loss_section']:.2f} bwd {p['backward_sweeps']:.2f} tail {p['tail']:.2f}  sweeps " + " ".join(f"{x:.2f}" for x in p["sweeps_in_launch_order"]))
PY
