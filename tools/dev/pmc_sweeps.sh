#!/bin/bash
# HBM traffic of the stand-alone sweeps under different PGASR_LSTM_FLAGS (run ON the GPU box):  bash tools/dev/pmc_sweeps.sh "0 8"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for f in $1; do
  export PGASR_LSTM_FLAGS=$f
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcsw_${f}_$c
    PGASR_LSTM_FLAGS=$f timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmcsw_${f}_$c -- python3 $R/tools/dev/tools_sweep_once.py > $O/pmcsw_${f}_$c.log 2>&1
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$O/pmcsw_${f}_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "$c" and "lstm_" in r["Kernel_Name"]:
            acc["fwd" if "fwd" in r["Kernel_Name"] else "bwd"][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, v in acc.items():
    vals = sorted(v.values())
    print("flags=$f $c", k, "launches", len(vals), "median KB", round(vals[len(vals)//2]), "min", round(vals[0]), "max", round(vals[-1]))
PY
  done
done
