"""Turn the rocprofv3 outputs of tools/profile_round5.sh into the committed summaries under profiles/:
  r05_f32_kernel_stats.csv, r05_bf16x3_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py [--precision bf16x3]`
  r05_pmc.json        HBM traffic per launch of the FED (and, backward, streamed) sweeps of either precision mode
  r05_gemm_pmc.json / .txt   SQ / TCC counters of the six-product GEMM kernels + the chip's sustained bare-MFMA rate
  r05_mfma_peak.txt   tools/mfma_peak.bin
HBM bytes follow MI355X_MICROARCH.md's HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE in separate passes, unit KB, and on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag = "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
KEYS = ("lstm_fwd_kernel", "lstm_bwd_kernel", "gemm_x6c_kernel", "gemm_t6_kernel", "gemm_x3w_kernel", "gemm_x3c_kernel", "gemm_t256_kernel")


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


os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
doc = {"source": "tools/profile_round5.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean per launch, counter unit KB; "
                 "the FED (backward: and streamed) sweeps of the timed step with their producer GEMM run FIRST and to completion "
                 "(a counter pass serialises kernels; tools/dev/tools_fed_sweep.py with PREC=<mode>)",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> "
                     "hbm_read_bytes = 2*FETCH_SIZE*1024; WRITE_SIZE*1024 as is",
       "algorithmic_bytes_per_launch": {"lstm_fwd_kernel": "262 MB read + 392 MB written", "lstm_bwd_kernel": "392 MB read + 262 MB written"}}
for prec in ("f32", "bf16x3"):
    t = traffic(per_dispatch(f"fedfetch_{prec}", "FETCH_SIZE"), per_dispatch(f"fedwrite_{prec}", "WRITE_SIZE"))
    doc[prec] = {"kernels": {k: v for k, v in t.items() if k.startswith("lstm_") or k.startswith("gemm_")}}
    st = newest(f"{tag}_stats_{prec}/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(root, "profiles", f"{tag}_{prec}_kernel_stats.csv"))
# round 5: the loss section's kernels (HBM GB/s for A4 / A5 / A9 / A12), stand-alone at the headline shape
LOSS_KEYS = {"head_logsoftmax_kernel": 65.5e6 + 2 * 3.7e6, "ctc_lattice_kernel": None, "ctc_grad_kernel": 2 * 3.7e6, "frame_argmax_sample_kernel": 3.7e6 + 0.26e6,
             "ctc_collapse_kernel": None, "edit_distance_kernel": None, "beam_small_kernel": 3.7e6}


def per_dispatch_any(dirname, counter, keys):
    f = newest(f"{tag}_{dirname}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    if f:
        for r in csv.DictReader(open(f)):
            k = next((key for key in keys if key in r["Kernel_Name"]), None)
            if r["Counter_Name"] == counter and k:
                acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}


lt = traffic(per_dispatch_any("lossfetch", "FETCH_SIZE", LOSS_KEYS), per_dispatch_any("losswrite", "WRITE_SIZE", LOSS_KEYS))
tl = newest("loss_kernels_time.log")
times = {}
if tl:
    for line in open(tl):
        if line.startswith("{"):
            times = json.loads(line).get("us_per_launch", {})
name_of = {"head_logsoftmax_kernel": "head_logsoftmax", "ctc_lattice_kernel": "ctc_lattice", "ctc_grad_kernel": "ctc_grad_from_lattice",
           "frame_argmax_sample_kernel": "frame_argmax_sample", "ctc_collapse_kernel": "ctc_collapse", "edit_distance_kernel": "edit_distance"}
for k, v in lt.items():
    v["algorithmic_bytes_per_launch"] = LOSS_KEYS.get(k)
    us = times.get(name_of.get(k, ""))
    if us:
        v["us_per_launch_unprofiled"] = us
        v["hbm_GBps"] = round(v["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
doc["loss_section"] = {"source": "tools/dev/r5_loss_kernels.py under --pmc FETCH_SIZE / WRITE_SIZE (T=1000, B=32, V=29, L=100, K=512); beam_small_kernel is "
                                 "launched twice per repetition (beam 16 and beam 5: the mean is over both); times from the same tool un-profiled",
                       "kernels": lt, "us_per_launch_unprofiled": times}
json.dump(doc, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps({p: {k: v["hbm_bytes_per_launch"] for k, v in doc[p]["kernels"].items()} for p in ("f32", "bf16x3")}, indent=1))

# GEMM counters
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
seen = {}
for f in glob.glob(os.path.join(out, f"{tag}_prof_*", f"{tag}_gemm6_pmc*", "**", "*counter_collection.csv"), recursive=True):
    key = re.search(rf"{tag}_gemm6_pmc\d+", f).group(0)
    if key not in seen or os.path.getmtime(f) > os.path.getmtime(seen[key]):
        seen[key] = f
for f in seen.values():
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and k.startswith("gemm_"):
            acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
lines = ["# mean per launch; source: tools/profile_round5.sh on QUICK=1 tools/dev/tools_gemm6.py (x6c: M=32000, N=2048 K=512 and N=512 K=2048 averaged; t6: dW_ih 2048x512x32000 split-K 16)"]
kern = {}
for k in sorted(acc):
    c = {n: sum(v.values()) / len(v) for n, v in acc[k].items()}
    lines.append(k)
    for n in sorted(c):
        lines.append(f"    {n:34s} {c[n]:16.0f}")
    rec = {}
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        rec["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)     # GRBM_GUI_ACTIVE is summed over the 8 XCDs, MFMA busy cycles over the 1024 SIMDs
        lines.append(f"    -> MFMA pipe busy (per SIMD, of kernel cycles)   {rec['mfma_busy']:.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        rec["wait_any"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]; rec["wait_inst"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        rec["active"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        lines.append(f"    -> wave cycles: parked (s_waitcnt / barrier) {rec['wait_any']:.3f}  issue-stalled {rec['wait_inst']:.3f}  issuing {rec['active']:.3f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        lines.append(f"    -> LDS bank-conflict cycles / active   {rec['lds_bank_conflict']:.3f}")
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        rec["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        lines.append(f"    -> L2 hit rate   {rec['l2_hit']:.3f}")
    kern[k] = rec
sustained = None
mp = newest("mfma_peak.txt")
if mp:
    shutil.copy(mp, os.path.join(root, "profiles", f"{tag}_mfma_peak.txt"))
    vals = [float(m.group(1)) for m in re.finditer(r"ms\s+(\d+) TF", open(mp).read())]
    sustained = max(vals) if vals else None
json.dump({"source": f"profiles/{tag}_gemm_pmc.txt (rocprofv3 --pmc, tools/profile_round5.sh); sustained rate: profiles/{tag}_mfma_peak.txt (tools/mfma_peak.hip)",
           "sustained_bare_mfma_tflops": sustained, "kernels": kern}, open(os.path.join(root, "profiles", f"{tag}_gemm_pmc.json"), "w"), indent=1)
open(os.path.join(root, "profiles", f"{tag}_gemm_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
