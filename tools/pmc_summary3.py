"""Turn the rocprofv3 outputs of tools/profile_round3.sh into the committed summaries under profiles/.

  python tools/pmc_summary3.py r03      (reads gpurun_out/r03_prof_*/, writes profiles/r03_bench_kernel_stats.csv, r03_pmc.json, r03_gemm_pmc.txt)

HBM bytes follow MI355X_MICROARCH.md's HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE are collected in separate passes, the unit
is KB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 is taken as is."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
KEYS = ("lstm_fwd_kernel", "lstm_bwd_kernel", "gemm_x3w_kernel", "gemm_x3c_kernel", "gemm_t256_kernel", "gemm_x3w256_kernel", "gemm_bf16x3_kernel",
        "gemm_f32_kernel", "gemm_reduce_kernel", "ctc_lattice_kernel", "ctc_grad_kernel", "edit_distance_kernel", "beam_small_kernel", "stream_copy_kernel")


def newest(pattern):
    hits = glob.glob(os.path.join(out, f"{tag}_prof_*", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    for key in KEYS:
        if key in name:
            return key
    return None


def per_dispatch(dirname, counter):
    f = newest(f"{tag}_{dirname}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if r["Counter_Name"] == counter and k:
                acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}


def traffic(fetch, write):
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": max(nf, nw), "hbm_read_bytes_per_launch": int(2 * f * 1024),
                      "hbm_write_bytes_per_launch": int(w * 1024), "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
    return kernels


fed = traffic(per_dispatch("fedfetch", "FETCH_SIZE"), per_dispatch("fedwrite", "WRITE_SIZE"))
seq = traffic(per_dispatch("fetch", "FETCH_SIZE"), per_dispatch("write", "WRITE_SIZE"))
doc = {"source": "tools/profile_round3.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean per launch, counter unit KB",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> "
                     "hbm_read_bytes = 2*FETCH_SIZE*1024; WRITE_SIZE*1024 as is",
       "fed_sweeps": {"what": "the FED forward / backward sweeps of the timed step (pgasr_lstm_layer_fwd_fed / _bwd_fed: helpers poll the tile counters, "
                              "read the fed rows with agent-scope loads, backward helpers apply the dropout mask) with their producer GEMM "
                              "(pgasr_gemm_x3w_feed_f32) run FIRST and to completion -- a counter pass serialises kernels, the side-by-side order "
                              "cannot be profiled (tools/dev/tools_fed_sweep.py)",
                      "kernels": {k: v for k, v in fed.items() if k.startswith("lstm_") or k.startswith("gemm_x3")}},
       "whole_step_sequential_order": {"what": "`PGASR_ALLOW_SEQUENTIAL=1 python3 bench.py --steps 3 --warmup 1`: every kernel of the step, sweeps in their "
                                               "un-fed form (projection before sweep)", "kernels": seq},
       "kernels": {**seq, **{k: v for k, v in fed.items() if k.startswith("lstm_")}}}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(doc, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
st = newest(f"{tag}_stats/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(root, "profiles", f"{tag}_bench_kernel_stats.csv"))
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in doc["kernels"].items()}, indent=1))

# GEMM counters
lines = [f"# mean per launch; source: tools/profile_round3.sh {tag} on tools/dev/tools_gemm3.py (M=32000: NT K=512 N=2048, NN K=2048 N=512, TN dW_ih 2048x512x32000 split-K 8 and 16)",
         "# default kernels: gemm_x3c_kernel (plain x3w), gemm_t256_kernel (weight gradients); gemm_x3w_kernel = the 256x128 kernel the feeds use (PGASR_X3W_TILE=128 passes)"]
for prefix in ("gemm_pmc", "gemm128_pmc"):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    seen = {}
    for f in glob.glob(os.path.join(out, f"{tag}_prof_*", f"{tag}_{prefix}*", "**", "*counter_collection.csv"), recursive=True):
        d = os.path.basename(f.split(os.sep + f"{tag}_{prefix}")[0]) + prefix + f.split(f"{tag}_{prefix}")[1].split(os.sep)[0]
        key = f"{tag}_{prefix}" + f.split(f"{tag}_{prefix}")[1].split(os.sep)[0]
        if key not in seen or os.path.getmtime(f) > os.path.getmtime(seen[key]):
            seen[key] = f
    for f in sorted(seen.values()):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k or "reduce" in k:
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k in sorted(acc):
        lines.append(k)
        c = {n: sum(v.values()) / len(v) for n, v in acc[k].items()}
        for n in sorted(c):
            lines.append(f"    {n:28s} {c[n]:16.0f}")
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            lines.append(f"    -> MFMA pipe busy (per SIMD, of kernel cycles)  {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
            lines.append(f"    -> VALU issue (4 cycles/inst, per SIMD)         {4 * c.get('SQ_INSTS_VALU', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
        if c.get("SQ_WAVE_CYCLES"):
            lines.append(f"    -> wave cycles waiting (SQ_WAIT_ANY)   {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
            lines.append(f"    -> wave cycles in issue stalls          {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
            lines.append(f"    -> wave cycles issuing (ACTIVE_INST)   {c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"    -> LDS bank-conflict cycles / active   {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
        if (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            lines.append(f"    -> L2 hit rate                          {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
open(os.path.join(root, "profiles", f"{tag}_gemm_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:6]))
