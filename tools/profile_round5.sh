#!/bin/bash
# Profiling recipe of round 5 (run ON the GPU box from the repo root):  bash tools/profile_round5.sh
#   1. rocprofv3 --kernel-trace --stats on the default bench command (precision f32) and on --precision bf16x3
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) on the FED sweeps of either mode, producer GEMM run first (tools/dev/tools_fed_sweep.py)
#   3. SQ counters on the six-product GEMM kernels at the path's shapes (tools/dev/tools_gemm6.py)
#   4. the chip's sustained bare-MFMA rate (tools/mfma_peak.bin)
#   5. (new in round 5) FETCH_SIZE / WRITE_SIZE on the loss section's kernels -- CTC head, CTC lattice + gradient, arg-max + sampler,
#      collapse, edit distance, the beam-16 reward hypothesis -- stand-alone at the headline shape (tools/dev/r5_loss_kernels.py)
# tools/pmc_summary5.py turns the outputs into profiles/r05_*.
set -u
TAG=r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
RUN=$(date +%H%M%S)
O=$R/gpurun_out/${TAG}_prof_$RUN
mkdir -p "$O"
[ -x "$R/tools/mfma_peak.bin" ] && timeout -k 5 60 "$R/tools/mfma_peak.bin" > "$O/mfma_peak.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for prec in f32 bf16x3; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats_$prec" -- python3 "$R/bench.py" --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > "$O/stats_$prec.log" 2>&1 || echo "stats pass $prec failed"
  echo "stats $prec done"
  PREC=$prec timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${TAG}_fedfetch_$prec" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$O/fedfetch_$prec.log" 2>&1 || echo "fed fetch pass $prec failed"
  PREC=$prec timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${TAG}_fedwrite_$prec" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$O/fedwrite_$prec.log" 2>&1 || echo "fed write pass $prec failed"
  echo "traffic $prec done"
done
timeout -k 10 120 python3 "$R/tools/dev/r5_loss_kernels.py" > "$O/loss_kernels_time.log" 2>&1 || echo "loss kernels timing failed"
REPS=3 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${TAG}_lossfetch" -- python3 "$R/tools/dev/r5_loss_kernels.py" > "$O/lossfetch.log" 2>&1 || echo "loss fetch pass failed"
REPS=3 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${TAG}_losswrite" -- python3 "$R/tools/dev/r5_loss_kernels.py" > "$O/losswrite.log" 2>&1 || echo "loss write pass failed"
echo "loss section traffic done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  QUICK=1 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_gemm6_pmc$i" -- python3 "$R/tools/dev/tools_gemm6.py" > "$O/gemm6_pmc$i.log" 2>&1 || echo "gemm6 pass $i failed"
  echo "gemm pmc $i done"
done
echo "profile_round5 done: $O"
