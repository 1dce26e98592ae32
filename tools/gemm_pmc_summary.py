"""Summarise tools/gemm_pmc.sh output into profiles/<tag>_gemm_pmc.txt (mean counter value per launch and kernel)."""
import csv, glob, os, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> counter -> dispatch -> sum
# gpurun merges runs into the same directories: keep the newest counter file of each pass
newest = {}
for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_gemm_pmc*", "**", "*counter_collection.csv"), recursive=True):
    d = f.split(os.sep + "gpurun_out" + os.sep)[1].split(os.sep)[0]
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in sorted(newest.values()):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
lines = [f"# mean per launch; source: tools/gemm_pmc.sh {tag} on tools/dev/tools_gemm2.py (M=32000: NT K=512 N=2048, NN K=2048 N=512)"]
for k in sorted(acc):
    lines.append(k)
    c = {n: sum(v.values()) / len(v) for n, v in acc[k].items()}
    for n in sorted(c):
        lines.append(f"    {n:28s} {c[n]:16.0f}")
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; MFMA busy cycles are summed over the 1024 SIMDs
        lines.append(f"    -> MFMA pipe busy (per SIMD, of kernel cycles)  {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
        lines.append(f"    -> VALU issue (4 cycles/inst, per SIMD)         {4 * c.get('SQ_INSTS_VALU', 0) / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        lines.append(f"    -> wave cycles waiting (SQ_WAIT_ANY)   {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
        lines.append(f"    -> wave cycles issuing (ACTIVE_INST)   {c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"    -> LDS bank-conflict cycles / active   {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        lines.append(f"    -> L2 hit rate                          {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
out = os.path.join(root, "profiles", f"{tag}_gemm_pmc.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
