#!/bin/bash
# Profiling recipe of round 3 (run ON the GPU box from the repo root):  bash tools/profile_round3.sh r03
#   1. rocprofv3 --kernel-trace --stats on the default bench command  -> <tag>_stats (average launch durations)
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) on the FED sweeps, producer GEMM run first (tools/dev/tools_fed_sweep.py)
#   3. FETCH_SIZE / WRITE_SIZE on the whole step in the sequential order (the other kernels' traffic)
#   4. SQ / TCC counters on the GEMM kernels of the path's shapes (tools/dev/tools_gemm3.py)
# Every pass writes into a directory of its own under gpurun_out/ (a new one per call: nothing is deleted);
# tools/pmc_summary3.py takes the newest output of each pass and writes profiles/<tag>_*.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
RUN=$(date +%H%M%S)
O=$R/gpurun_out/${TAG}_prof_$RUN
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > "$O/stats.log" 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${TAG}_fedfetch" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$O/fedfetch.log" 2>&1 || echo "fed fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${TAG}_fedwrite" -- python3 "$R/tools/dev/tools_fed_sweep.py" > "$O/fedwrite.log" 2>&1 || echo "fed write pass failed"
export PGASR_ALLOW_SEQUENTIAL=1   # counter passes serialise kernels: the feed-ahead paths fall back to the sequential order (same kernels)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${TAG}_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > "$O/fetch.log" 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${TAG}_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 > "$O/write.log" 2>&1 || echo "write pass failed"
unset PGASR_ALLOW_SEQUENTIAL
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  REPS=2 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_gemm_pmc$i" -- python3 "$R/tools/dev/tools_gemm3.py" > "$O/gemm_pmc$i.log" 2>&1 || echo "gemm pass $i failed"
  REPS=2 WHICH=nt,nn PGASR_X3W_TILE=128 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$O/${TAG}_gemm128_pmc$i" -- python3 "$R/tools/dev/tools_gemm3.py" > "$O/gemm128_pmc$i.log" 2>&1 || echo "gemm128 pass $i failed"
done
echo "profile_round3 done: $O"
