#!/bin/bash
# PMC passes over the GEMM kernels on the path's two big shapes (run ON the GPU box from the repo root):
#   bash tools/gemm_pmc.sh r01 ; python tools/gemm_pmc_summary.py r01   -> profiles/r01_gemm_pmc.txt
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf "$O/${TAG}_gemm_pmc$i"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_gemm_pmc$i" -- python3 "$R/tools/dev/tools_gemm2.py" > "$O/${TAG}_gemm_pmc$i.log" 2>&1 || { echo "pass $i failed"; exit 1; }
done
echo "gemm_pmc done"
