#!/bin/bash
# round 4: the six-product kernels -- variants (PGASR_X6_VAR), the no-SLP build, PMC passes; then the f32 step end to end
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
for lib in libpgasr_hip.so libpgasr_hip_noslp.so; do
  for v in 0 1 2; do
    echo "== $lib VAR=$v" >> $O/ab.log
    PGASR_HIP_LIB=$R/policy_gradient_asr_amd/$lib PGASR_X6_VAR=$v QUICK=1 timeout -k 10 120 python3 tools/dev/tools_gemm6.py >> $O/ab.log 2>&1 || echo "failed" >> $O/ab.log
  done
done
for v in 0 1; do
  echo "== f32 step VAR=$v" >> $O/ab.log
  PGASR_X6_VAR=$v PREC=f32 STEPS=30 timeout -k 10 200 python3 tools/dev/tools_precision_phases.py >> $O/ab.log 2>&1 || echo "failed" >> $O/ab.log
done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  PGASR_X6_VAR=${PMC_VAR:-0} QUICK=1 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/pmc$i" -- python3 "$R/tools/dev/tools_gemm6.py" > "$O/pmc$i.log" 2>&1 || echo "pass $i failed" >> $O/ab.log
done
python3 - <<PY >> $O/ab.log
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob('$O/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm' not in k: continue
        k=k.replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        acc[k][r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
for k in sorted(acc):
    print(k)
    c={n: sum(v.values())/len(v) for n,v in acc[k].items()}
    for n in sorted(c): print(f'    {n:34s} {c[n]:16.0f}')
    if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print('    -> MFMA busy', round(c['SQ_VALU_MFMA_BUSY_CYCLES']/(c['GRBM_GUI_ACTIVE']/8*1024),3))
    if c.get('SQ_WAVE_CYCLES'):
        print('    -> wait_any', round(c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES'],3), 'wait_inst', round(c.get('SQ_WAIT_INST_ANY',0)/c['SQ_WAVE_CYCLES'],3), 'active', round(c.get('SQ_ACTIVE_INST_ANY',0)/c['SQ_WAVE_CYCLES'],3))
PY
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
tail -n 100 $O/ab.log
