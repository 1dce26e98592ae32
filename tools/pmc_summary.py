"""Turn the rocprofv3 outputs of tools/profile_round.sh into the committed summaries under profiles/.

  python tools/pmc_summary.py r01        (reads gpurun_out/r01_{stats,fetch,write}/, writes profiles/r01_*)

HBM bytes follow MI355X_MICROARCH.md's HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE are collected in
separate passes, the unit is KB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is taken as is."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def one(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return max(hits, key=os.path.getmtime)      # gpurun merges runs into the same directory: take the newest


def short(name):
    for key in ("lstm_fwd_kernel", "lstm_bwd_kernel", "gemm_x3w_kernel", "gemm_bf16x3_kernel", "gemm_f32_kernel", "gemm_reduce_kernel",
                "ctc_lattice_kernel", "ctc_grad_kernel", "edit_distance_kernel", "beam_small_kernel", "stream_copy_kernel"):
        if key in name:
            return key
    return None


def counter_means(dirname, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(one(f"{tag}_{dirname}/**/*counter_collection.csv"))):
        if r["Counter_Name"] == counter and short(r["Kernel_Name"]):
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    # rocprofv3 emits one row per (dispatch, dimension instance): sum per dispatch first
    return acc


def per_dispatch(dirname, counter):
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(one(f"{tag}_{dirname}/**/*counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] == counter and k:
            acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}


fetch = per_dispatch("fetch", "FETCH_SIZE")
write = per_dispatch("write", "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    kernels[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": max(nf, nw),
                  "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
doc = {"source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                 "`PGASR_ALLOW_SEQUENTIAL=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity` (the profiler serialises kernels: sequential order of the same kernels); mean per launch; counter unit KB",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM "
                     "section) -> hbm_read_bytes = 2*FETCH_SIZE*1024; WRITE_SIZE*1024 as is",
       "kernels": kernels}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(doc, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
shutil.copy(one(f"{tag}_stats/**/*kernel_stats.csv"), os.path.join(root, "profiles", f"{tag}_bench_kernel_stats.csv"))
print(json.dumps(kernels, indent=1))
