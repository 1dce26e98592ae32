#!/bin/bash
# one-box diagnosis of the 256 x 256 tiles: A/B against the 128-wide kernels, the no-DMA / no-MFMA variants, PMC
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
echo "== default (x3w 128 tile, TN 256 tile)";          CHECK=1 python tools/dev/tools_gemm3.py
echo "== TN 128 tile";                                   WHICH=tn PGASR_TN_TILE=128 python tools/dev/tools_gemm3.py
echo "== x3w 256 tile";                                  WHICH=nt,nn PGASR_X3W_TILE=256 python tools/dev/tools_gemm3.py
echo "== x3w 256 tile, no DMA in the k-loop";            WHICH=nt,nn PGASR_X3W_TILE=256 PGASR_X3W_DIAG=1 python tools/dev/tools_gemm3.py
echo "== x3w 256 tile, no MFMA";                         WHICH=nt,nn PGASR_X3W_TILE=256 PGASR_X3W_DIAG=2 python tools/dev/tools_gemm3.py
echo "== x3w 256 tile, neither";                         WHICH=nt,nn PGASR_X3W_TILE=256 PGASR_X3W_DIAG=3 python tools/dev/tools_gemm3.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/r3_counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  REPS=2 PGASR_X3W_TILE=256 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/r03_gemm_pmc$i" -- python3 "$R/tools/dev/tools_gemm3.py" > "$O/r03_gemm_pmc$i.log" 2>&1 || echo "pass $i failed"
done
echo "diag done"
