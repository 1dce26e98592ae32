#!/bin/bash
# PMC passes over the x3w kernel selected by PGASR_X3W_TILE (default c) on the path's two shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r03c}
export PGASR_X3W_TILE=${PGASR_X3W_TILE:-c}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  WHICH=nt,nn REPS=2 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_pmc$i" -- python3 "$R/tools/dev/tools_gemm3.py" > "$O/${TAG}_pmc$i.log" 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob('$O/${TAG}_pmc*/**/*counter_collection.csv', recursive=True):
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
