#!/bin/bash
# Profiling recipe for one round (run ON the GPU box from the repo root):
#   bash tools/profile_round.sh r01
# Writes gpurun_out/<tag>_{stats,fetch,write}/ ; tools/pmc_summary.py turns them into profiles/<tag>_*.
# PMC passes are separate runs and carry no trace domains besides what --pmc implies.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export PGASR_ALLOW_SEQUENTIAL=1   # counter passes serialise kernels: the feed-ahead paths fall back to the sequential order (same kernels)
rm -rf "$O/${TAG}_stats" "$O/${TAG}_fetch" "$O/${TAG}_write"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-parity > "$O/${TAG}_stats.log" 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${TAG}_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity > "$O/${TAG}_fetch.log" 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${TAG}_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity > "$O/${TAG}_write.log" 2>&1
echo "profile_round rc=$?"
