#!/bin/bash
# SQ / TCC counters of the six-product GEMM kernels (tools/profile_round4.sh part 3 + 4 alone), into the newest r04_prof_* directory layout
set -u
TAG=r04; R=${GRAFT_REPO_ROOT:-$(pwd)}; RUN=$(date +%H%M%S); O=$R/gpurun_out/${TAG}_prof_$RUN; mkdir -p "$O"
[ -x "$R/tools/mfma_peak.bin" ] && timeout -k 5 60 "$R/tools/mfma_peak.bin" > "$O/mfma_peak.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  QUICK=1 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_gemm6_pmc$i" -- python3 "$R/tools/dev/tools_gemm6.py" > "$O/gemm6_pmc$i.log" 2>&1 || echo "gemm6 pass $i failed"
  echo "gemm pmc $i done"
done
cd $R && QUICK=1 python3 tools/dev/tools_gemm6.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm6_plain.txt
