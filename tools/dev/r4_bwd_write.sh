#!/bin/bash
# WRITE_SIZE of the backward sweep: plain against streamed (slab publication + the flusher's L2 write-backs), f32, stand-alone
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in plain streamed_alone; do
  MODES=$m REPS=2 PREC=f32 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/bw_$m" -- python3 "$R/tools/dev/tools_streamed_bwd.py" > "$O/bw_$m.log" 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob('$O/bw_$m/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lstm_bwd_kernel' in r['Kernel_Name']: acc[r['Dispatch_Id']]+=float(r['Counter_Value'])
print('$m', 'WRITE_SIZE per backward launch MB', [round(v*1024/1e6) for v in acc.values()])
PY
  rm -rf "$O/bw_$m"
done
