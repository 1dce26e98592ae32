#!/bin/bash
# kernel timeline of the last train step of a short bench run (development aid):  bash tools/dev/tools_trace_step.sh [min_us] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_step_$(date +%H%M%S)
mkdir -p "$O"
MIN=${1:-14}; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 "$R/bench.py" --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-extra-legs --long-steps 0 "$@" > "$O/log.txt" 2>&1
python3 "$R/tools/step_timeline.py" "$O" $MIN
