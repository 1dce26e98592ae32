#!/bin/bash
# round 4: non-temporal result stores in the sweeps (libpgasr_hip_nt.so) against plain ones: stand-alone sweeps, HBM write traffic, the steps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/nt.log
for lib in libpgasr_hip.so libpgasr_hip_nt.so; do
  echo "== $lib" >> $O/nt.log
  for p in bf16x3 f32; do PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=$p timeout -k 10 120 python3 tools/dev/tools_sweep_once.py 2>&1 | grep flags >> $O/nt.log; done
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/nt.log
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=bf16x3 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/nt.log
done
cd /tmp && export TMPDIR=/tmp
for lib in libpgasr_hip.so libpgasr_hip_nt.so; do
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/ntw_$lib" -- python3 "$R/tools/dev/tools_sweep_once.py" > "$O/ntw_$lib.log" 2>&1
  python3 - <<PY >> $O/nt.log
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$O/ntw_$lib/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'lstm_' in k and 'prepare' not in k: acc[k.split('(')[0][-22:]][r['Dispatch_Id']]+=float(r['Counter_Value'])
for k,v in acc.items(): print('$lib WRITE_SIZE per launch MB', k, sum(v.values())/len(v)*1024/1e6)
PY
  rm -rf "$O/ntw_$lib"
done
cd $R; python3 - <<PY
import json
for l in open("$O/nt.log"):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    p=d["phases"]; print(f"   {d['precision']} {d['ms_per_step']:.2f} ms  fwd {p['forward_sweeps']:.2f} bwd {p['backward_sweeps']:.2f} tail {p['tail']:.2f}  sweeps " + " ".join(f"{x:.2f}" for x in p["sweeps_in_launch_order"]))
PY
