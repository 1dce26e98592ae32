#!/bin/bash
# kernel timeline of one streamed backward sweep with its gated weight-gradient launch (development aid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_streamed_$(date +%H%M%S)
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
MODES=streamed+dW REPS=3 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 "$R/tools/dev/tools_streamed_bwd.py" > "$O/log.txt" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last sweep
idx = [i for i, r in enumerate(rows) if "lstm_bwd_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx - 3: idx + 12]:
    print(f'{r["Kernel_Name"][:70]:70s} start {(int(r["Start_Timestamp"]) - t0) / 1000:9.1f} us  end {(int(r["End_Timestamp"]) - t0) / 1000:9.1f} us  grid {r.get("Grid_Size", "")}')
PY
