#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
rm -f $O/ring2.log
timeout -k 10 600 python3 -m pytest tests/test_dense_lstm_gpu.py -x -q -k "blstm or streamed or encoder or seq2seq or sweep_error" > $O/t4.log 2>&1; echo "pytest rc=$?" >> $O/t4.log; tail -n 3 $O/t4.log
for lib in libpgasr_hip.so libpgasr_hip_f12.so; do
  echo "== $lib" >> $O/ring2.log
  for rep in 1 2; do for p in bf16x3 f32; do PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=$p timeout -k 10 120 python3 tools/dev/tools_sweep_once.py 2>&1 | grep flags >> $O/ring2.log; done; done
done
cd /tmp && export TMPDIR=/tmp
for lib in libpgasr_hip.so libpgasr_hip_f12.so; do
  PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PREC=f32 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/rw_$lib" -- python3 "$R/tools/dev/tools_sweep_once.py" > "$O/rw_$lib.log" 2>&1
  python3 - <<PY >> $O/ring2.log
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
cat $O/ring2.log
