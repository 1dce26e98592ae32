#!/bin/bash
# round 4: backward staging ring of 16 (default) / 12 / 10 steps: stand-alone sweep time, HBM write traffic, the f32 step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/ring.log
for lib in libpgasr_hip.so libpgasr_hip_r12.so libpgasr_hip_r10.so; do
  echo "== $lib" >> $O/ring.log
  for p in bf16x3 f32; do PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=$p timeout -k 10 120 python3 tools/dev/tools_sweep_once.py 2>&1 | grep flags >> $O/ring.log; done
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 STEPS=40 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py 2>&1 | grep -v amdgpu.ids >> $O/ring.log
done
cd /tmp && export TMPDIR=/tmp
for lib in libpgasr_hip.so libpgasr_hip_r12.so libpgasr_hip_r10.so; do
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/rw_$lib" -- python3 "$R/tools/dev/tools_sweep_once.py" > "$O/rw_$lib.log" 2>&1
  python3 - <<PY >> $O/ring.log
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$O/rw_$lib/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'lstm_fwd_kernel' in k or 'lstm_bwd_kernel' in k: acc['fwd' if 'fwd' in k else 'bwd'][r['Dispatch_Id']]+=float(r['Counter_Value'])
for k,v in acc.items(): print('$lib WRITE_SIZE per launch MB', k, round(sum(v.values())/len(v)*1024/1e6))
PY
  rm -rf "$O/rw_$lib"
done
cd $R; python3 - <<PY
import json
for l in open("$O/ring.log"):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    p=d["phases"]; print(f"   {d['precision']} {d['ms_per_step']:.2f} ms  fwd {p['forward_sweeps']:.2f} bwd {p['backward_sweeps']:.2f} tail {p['tail']:.2f}  sweeps " + " ".join(f"{x:.2f}" for x in p["sweeps_in_launch_order"]))
PY
